#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/ab_attention_ablate.py > gpurun_out/r03_attn_ablate.log 2>&1; echo "rc=$?"; cat gpurun_out/r03_attn_ablate.log
