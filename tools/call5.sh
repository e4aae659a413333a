mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null
echo "##### attention tests (all variants, prescaled)"
timeout 900 python -m pytest tests/test_unet_ops.py -k "attention" -q -p no:cacheprovider --tb=short 2>&1 | grep -v "^E    \+ " | tail -30 | cut -c1-400 | tee gpurun_out/attn3_tests.log
echo "##### attention A/B"
timeout 300 python tools/ab_attention.py 2>&1 | grep -v "d= 80\|d=160\|d= 64" | tail -30 | tee gpurun_out/ab_attention.log
echo "##### unet tests (prescaled to_q in the engine)"
timeout 1500 python -m pytest tests/test_unet.py tests/test_pipeline_mixin.py tests/test_controlnet.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | grep -v "^E    \+ " | tail -15 | cut -c1-300 | tee gpurun_out/unet_tests.log
echo "##### bench"
timeout 600 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.log | tail -1 | cut -c1-1500
