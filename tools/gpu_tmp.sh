#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/ab_swizzle.py 8 > gpurun_out/r03_ab_swizzle_b8.log 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/r03_ab_swizzle_b8.log | tail -18
