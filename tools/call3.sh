mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null
export MVE_RUN_PENDING=1
echo "##### attention k_attention3 tests"
timeout 600 python -m pytest tests/test_unet_ops.py -k "d40_transposed" -q -p no:cacheprovider --tb=short 2>&1 | grep -v "^E    \+ " | tail -30 | cut -c1-400 | tee gpurun_out/attn3_tests.log
echo "##### attention A/B"
timeout 300 python tools/ab_attention.py 2>&1 | tail -24 | tee gpurun_out/ab_attention.log
echo "##### retest"
timeout 600 python -m pytest tests/test_lpips.py tests/test_mesh_loss.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | grep -v "^E    \+ " | tail -12 | cut -c1-300 | tee gpurun_out/retest2.log
echo "##### rocprof kernel trace of the first-run kernels"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_pending -o pending -- python -m pytest $REPO/tests/test_recon_loss.py $REPO/tests/test_mesh_reg.py $REPO/tests/test_lpips.py $REPO/tests/test_tonemapping.py $REPO/tests/test_blur.py $REPO/tests/test_shencoder.py $REPO/tests/test_mesh_grad.py $REPO/tests/test_mesh_loss.py -q -m gpu -p no:cacheprovider > $REPO/gpurun_out/prof_pending.log 2>&1
tail -3 $REPO/gpurun_out/prof_pending.log
cd $REPO
python tools/summarize_prof.py gpurun_out prof_pending > gpurun_out/prof_pending_summary.txt 2>&1; head -70 gpurun_out/prof_pending_summary.txt
find gpurun_out/prof_pending -name "*.db" -size +20M -delete
