#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/ab_deep_conv.py 64 > gpurun_out/r03_ab_deep_conv.log 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/r03_ab_deep_conv.log
