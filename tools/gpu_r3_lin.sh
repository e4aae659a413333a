#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/ab_linear_small.py 64 > gpurun_out/r03_ab_linear_small.log 2>&1; echo "rc=$?"; cat gpurun_out/r03_ab_linear_small.log
