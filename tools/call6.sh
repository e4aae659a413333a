mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null
echo "##### attention tests, verbose, first failure"
timeout 900 python -X faulthandler -m pytest tests/test_unet_ops.py -k "attention" -v -x -p no:cacheprovider --tb=short > gpurun_out/attn_tests_full.log 2>&1; grep -n "PASSED\|FAILED\|Fatal\|fault\|File \"/root/repo/tests\|Error" gpurun_out/attn_tests_full.log | cut -c1-200 | tail -40
echo "##### mip texture tests"
timeout 900 python -m pytest tests/test_mesh_ops.py tests/test_bake_ref.py tests/test_mesh_forward_ref.py -q -p no:cacheprovider --tb=short 2>&1 | grep -v "^E    \+ " | tail -40 | cut -c1-300 | tee gpurun_out/mip_tests.log
