#!/bin/bash
# Round 4, call 6: kernel trace of the nerf_optim iteration.
mkdir -p gpurun_out
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_optim -o optim -- python $REPO/tools/optim_profile.py 20 > $REPO/gpurun_out/prof_optim.log 2>&1
cd $REPO
python tools/summarize_prof.py gpurun_out prof_optim > gpurun_out/r04_optim_kernel_stats_v0.txt 2>&1
head -45 gpurun_out/r04_optim_kernel_stats_v0.txt | cut -c1-150
rm -rf gpurun_out/prof_optim
