#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/probe/probe3 > gpurun_out/r03_probe3_v2.log 2>&1; echo "probe3 rc=$?"; grep "^g_\|^kernel" gpurun_out/r03_probe3_v2.log | cut -c1-175
