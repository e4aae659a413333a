#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 700 python bench.py > gpurun_out/r05_bench_v2.log 2>&1; tail -c 400 gpurun_out/r05_bench_v2.log; echo
timeout 200 python tools/op_list.py 64 > gpurun_out/r05_oplist_pair.log 2>&1; tail -n 1 gpurun_out/r05_oplist_pair.log
MVE_RESIDUAL_PAIR=0 timeout 200 python tools/op_list.py 64 > gpurun_out/r05_oplist_plain.log 2>&1; tail -n 1 gpurun_out/r05_oplist_plain.log
timeout 200 python tools/scale_preview.py > gpurun_out/r05_scale_preview_v2.log 2>&1; grep -v amdgpu gpurun_out/r05_scale_preview_v2.log
timeout 200 python -m pytest tests/test_unet_ops.py -q -m gpu -x -k "pair_launches" 2>&1 | tail -n 3
