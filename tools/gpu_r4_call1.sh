#!/bin/bash
# Round 4, call 1: per-op list at 64 images with / without split-K, SMU power cap, baseline bench line of this round's box.
mkdir -p gpurun_out
rocm-smi --showmaxpower --showpower > gpurun_out/r04_smi_maxpower.log 2>&1
timeout 400 python tools/op_list.py 64 > gpurun_out/r04_oplist_default.log 2>&1; echo "oplist rc=$?"
MVE_GEMM_SPLITK=0 timeout 300 python tools/op_list.py 64 > gpurun_out/r04_oplist_nosplit.log 2>&1; echo "oplist nosplit rc=$?"
timeout 500 python bench.py --no-secondary 2>gpurun_out/r04_bench_v0.err | tee gpurun_out/r04_bench_v0.log | tail -1 | cut -c1-1200
tail -3 gpurun_out/r04_oplist_default.log; tail -1 gpurun_out/r04_oplist_nosplit.log; cat gpurun_out/r04_smi_maxpower.log | head -20
