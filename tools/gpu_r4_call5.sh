#!/bin/bash
# Round 4, call 5: where a nerf_optim iteration goes (synchronised parts + kernel trace); LayerNorm rows-per-wave check (norm tests + op list).
mkdir -p gpurun_out
REPO=$PWD
timeout 300 python tools/optim_profile.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_optim_profile_v0.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_optim -o optim -- python $REPO/tools/optim_profile.py 20 > $REPO/gpurun_out/prof_optim.log 2>&1
cd $REPO
python tools/summarize_prof.py gpurun_out 2>/dev/null | head -5
find gpurun_out/prof_optim -name "*stats*.csv" | head; f=$(find gpurun_out/prof_optim -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200 > gpurun_out/r04_optim_kernel_stats_v0.txt; cat gpurun_out/r04_optim_kernel_stats_v0.txt | head -45
rm -rf gpurun_out/prof_optim
timeout 600 python -m pytest tests/test_unet_ops.py -x -q -m gpu -k "layernorm or norms_read" 2>&1 | tail -3
timeout 300 python tools/op_list.py 64 2>&1 | grep "layernorm\|total" | awk '{s+=$(NF-2)} END{print}' | tail -1
timeout 300 python tools/op_list.py 64 2>&1 | tail -1
