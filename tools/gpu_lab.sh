#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/attn_lab/attn_lab ${1:-2} > gpurun_out/r03_attn_lab.log 2>&1; echo "rc=$?"; cat gpurun_out/r03_attn_lab.log
