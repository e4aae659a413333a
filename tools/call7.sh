mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null
echo "##### attention tests"
timeout 900 python -X faulthandler -m pytest tests/test_unet_ops.py -k "attention" -q -x -p no:cacheprovider --tb=short > gpurun_out/attn_tests_full.log 2>&1; grep -v "^E    \+ " gpurun_out/attn_tests_full.log | tail -15 | cut -c1-300
echo "##### attention A/B"
timeout 300 python tools/ab_attention.py 2>&1 | grep -v "d= 80\|d=160\|d= 64" | tail -34 | tee gpurun_out/ab_attention.log
echo "##### mip texture tests"
timeout 900 python -m pytest tests/test_mesh_ops.py -q -p no:cacheprovider --tb=short 2>&1 | grep -v "^E    \+ " | tail -30 | cut -c1-300 | tee gpurun_out/mip_tests.log
