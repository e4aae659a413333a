#!/bin/bash
for r in 0 1; do
  echo "MVE_GEMM_RED=$r"
  MVE_GEMM_RED=$r timeout 300 python tools/fwd_z123.py 2>&1 | grep "step ms"
  MVE_GEMM_RED=$r timeout 300 python tools/fwd_small.py 8 4 2>&1 | grep "forward ms"
done
