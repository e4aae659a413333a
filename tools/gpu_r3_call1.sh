#!/bin/bash
# Round 3, call 1: co-issue probe in cycles (tools/probe/probe3) + the round's starting bench line on this pool's boxes.
mkdir -p gpurun_out
timeout 300 tools/probe/probe3 > gpurun_out/r03_probe3.log 2>&1; echo "probe3 rc=$?"; tail -5 gpurun_out/r03_probe3.log
timeout 600 python bench.py 2>gpurun_out/r03_bench_v0.err | tee gpurun_out/r03_bench_v0.log | tail -1 | cut -c1-1500
