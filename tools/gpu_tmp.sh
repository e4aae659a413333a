#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --secondary-only 2>gpurun_out/r03_secondary.err | tee gpurun_out/r03_secondary.log | cut -c1-3000; tail -3 gpurun_out/r03_secondary.err
