#!/bin/bash
# MVE_GEMM_PP160_MINK sweep: Zero123++ step and the 8 / 16-image SD-1.5 forwards, same box
for k in 1440 2048 2880 4096 5760; do
  echo "MINK=$k"
  MVE_GEMM_PP160_MINK=$k timeout 300 python tools/fwd_z123.py 2>&1 | grep "step ms"
  MVE_GEMM_PP160_MINK=$k timeout 300 python tools/fwd_small.py 8 4 2>&1 | grep "forward ms"
  MVE_GEMM_PP160_MINK=$k timeout 300 python tools/fwd_small.py 16 4 2>&1 | grep "forward ms"
done
