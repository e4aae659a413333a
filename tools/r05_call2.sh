#!/bin/bash
# round 5, GPU call 2: bisect the pair mode at large batches (rows of a 64-image batch were ~100 % off), cross-kernel bitwise test of pair launches,
# the texture super-resolution composition on its own
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
L=gpurun_out/r05_pair_bisect.log
: > $L
timeout 300 python -m pytest tests/test_unet_ops.py -q -m gpu -x -s -k "pair_launches_round_identically" 2>&1 | grep -v "^$" | tail -n 25 >> $L
timeout 200 python -m pytest tests/test_unet.py -q -m gpu -x -k "enc_dec_equals" 2>&1 | tail -n 5 >> $L
for cfg in "" "MVE_UPSAMPLE_PHASES=0" "MVE_GEMM_STRICT_SPLITK=1" "MVE_DEBUG_TUNE=0" "MVE_GEMM_SPLITK=0" "MVE_RESIDUAL_PAIR=0" "MVE_GEMM_PP=0"; do
  env $cfg timeout 200 python tools/debug_pair_batch.py 64 16 2>&1 | grep "pair=" >> $L
done
cat $L
timeout 400 python tools/bench_parts.py texture_superres > gpurun_out/r05_texture_superres_v0.log 2>&1; tail -c 1500 gpurun_out/r05_texture_superres_v0.log
