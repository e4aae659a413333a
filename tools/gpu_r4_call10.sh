#!/bin/bash
# Round 4, call 10: GroupNorm with four rows in flight, k_xty chunks: tests + op list + the nerf_optim iteration.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_unet_ops.py tests/test_nerf.py -x -q -m gpu -k "norm or backward or fitting" 2>&1 | tail -3
timeout 300 python tools/op_list.py 64 > gpurun_out/r04_oplist_v3_gn_rows.log 2>&1; tail -1 gpurun_out/r04_oplist_v3_gn_rows.log
timeout 300 python tools/optim_profile.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_optim_profile_v2_xty.log
