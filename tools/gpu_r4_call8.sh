#!/bin/bash
# Round 4, call 8: NeRF backward with merged atomics (tests + iteration profile), then the rocprofv3 passes of the bench command.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_nerf.py tests/test_recon_loss.py -x -q -m gpu -k "backward or fitting or autograd or differentiable or nerf_optim" 2>&1 | tail -3
timeout 300 python tools/optim_profile.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_optim_profile_v1_merged_atomics.log
bash tools/profile_round.sh > gpurun_out/r04_profile_round.log 2>&1; tail -30 gpurun_out/r04_profile_round.log | cut -c1-200
du -sh gpurun_out/prof_* 2>/dev/null
