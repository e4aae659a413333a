#!/bin/bash
# Round 4, call 9: nerf_optim iteration with merged atomics (log), rocprofv3 passes of the bench command (summaries only come back).
mkdir -p gpurun_out
rm -rf gpurun_out/prof_* gpurun_out/pmca_* gpurun_out/pmc_a gpurun_out/pmc_b
timeout 300 python tools/optim_profile.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_optim_profile_v1_merged_atomics.log
PROF_TAG=r04 bash tools/profile_round.sh > gpurun_out/r04_profile_round.log 2>&1; tail -3 gpurun_out/r04_profile_round.log | cut -c1-200
cat gpurun_out/pmc_traffic.json | head -30; du -sh gpurun_out
