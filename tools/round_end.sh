#!/bin/bash
# Round-end measurement with the final binary: the default bench command, the rocprofv3 passes over it, the strong-scaling preview.
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_final.log 2>&1; tail -c 1500 gpurun_out/r04_bench_final.log; echo
PROF_TAG=r04 bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -5 gpurun_out/profile_round.log
python tools/scale_preview.py > gpurun_out/r04_scale_preview.log 2>&1; cat gpurun_out/r04_scale_preview.log | grep -v amdgpu
