mkdir -p gpurun_out
./tools/probe/probe2 2>&1 | tee gpurun_out/probe2.log
python -c 'import torch' 2>/dev/null
timeout 900 python -m pytest tests/test_mesh_ops.py -q -p no:cacheprovider --tb=short -k "mip" 2>&1 | grep -v "^E    \+ " | tail -12 | cut -c1-300 | tee gpurun_out/mip_tests.log
