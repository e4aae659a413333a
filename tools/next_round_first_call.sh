#!/bin/bash
# One GPU call that answers everything round 1 left open (~7 min of box time).  Logs land in gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/next_round_first_call.sh'
mkdir -p gpurun_out
python -m mvedit_amd.build > gpurun_out/build.log 2>&1 || tail -5 gpurun_out/build.log
echo "##### 1. pending tests (first run on hardware)"
bash tools/gpu_pending.sh
echo "##### 2. full verified suite"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_gpu.log | grep -v "where\|and  " | head -20
echo "##### 3. VAE per-op detail (which conv shapes sit below the class average)"
timeout 120 python tools/vae_check.py 8 --detail 2>&1 | tail -90 > gpurun_out/vae_detail.log; grep -E "^decode|^encode" gpurun_out/vae_detail.log
echo "##### 4. attention variants, same box"
timeout 120 python tools/ab_attention.py 2>&1 | tail -20 | tee gpurun_out/ab_attention.log
echo "##### 5. strong-scaling preview, eager vs graph replay"
timeout 300 python tools/scale_preview.py --graph 2>&1 | tail -12 | tee gpurun_out/scale_preview.log
echo "##### 6. bench"
timeout 600 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.log | tail -1 | cut -c1-600
