#!/bin/bash
MVE_GEMM_SMALL_BM=64 MVE_GEMM_SMALL_BM_MAX=100000 timeout 900 python -m pytest tests/test_slice_reduce.py tests/test_unet_ops.py -x -q -k "slice or gemm or conv or pair" 2>&1 | tail -2
for cfg in "0 0" "64 256" "64 512" "64 1024"; do
  set -- $cfg
  echo "SMALL_BM=$1 MAX=$2"
  MVE_GEMM_SMALL_BM=$1 MVE_GEMM_SMALL_BM_MAX=$2 timeout 300 python tools/fwd_z123.py 2>&1 | grep "step ms"
  MVE_GEMM_SMALL_BM=$1 MVE_GEMM_SMALL_BM_MAX=$2 timeout 300 python tools/fwd_small.py 8 4 2>&1 | grep "forward ms"
  MVE_GEMM_SMALL_BM=$1 MVE_GEMM_SMALL_BM_MAX=$2 timeout 300 python tools/fwd_small.py 16 4 2>&1 | grep "forward ms"
done
