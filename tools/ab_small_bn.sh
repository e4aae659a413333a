#!/bin/bash
for cfg in "64 256" "64 320" "64 384" "64 512" "0 256"; do
  set -- $cfg
  echo "SMALL_BN=$1 MAX=$2"
  MVE_GEMM_SMALL_BN=$1 MVE_GEMM_SMALL_BN_MAX=$2 timeout 300 python tools/fwd_z123.py 2>&1 | grep "step ms"
  MVE_GEMM_SMALL_BN=$1 MVE_GEMM_SMALL_BN_MAX=$2 timeout 300 python tools/fwd_small.py 8 4 2>&1 | grep "forward ms"
  MVE_GEMM_SMALL_BN=$1 MVE_GEMM_SMALL_BN_MAX=$2 timeout 300 python tools/fwd_small.py 16 4 2>&1 | grep "forward ms"
done
