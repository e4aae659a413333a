mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null
echo "##### attention k_attention3 tests"
timeout 900 python -m pytest tests/test_unet_ops.py -k "d40_transposed" -q -p no:cacheprovider --tb=short 2>&1 | grep -v "^E    \+ " | tail -30 | cut -c1-400 | tee gpurun_out/attn3_tests.log
echo "##### attention A/B"
timeout 300 python tools/ab_attention.py 2>&1 | tail -24 | tee gpurun_out/ab_attention.log
echo "##### attention PMC"
bash tools/pmc_attention.sh 0 8 11 2>&1 | tail -12
