#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/probe/probe3 > gpurun_out/r03_probe3_v3.log 2>&1; echo "probe3 rc=$?"; grep "^MFMA" gpurun_out/r03_probe3_v3.log
timeout 600 python -m pytest tests/test_unet_ops.py -m gpu -x -q -k "attention" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/ab_attention.py 2>&1 | grep -v amdgpu | head -14
