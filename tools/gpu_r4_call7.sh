#!/bin/bash
# Round 4, call 7: fewer K slices at the 8 x 8 level (op list), GEMM / conv kernel tests, LayerNorm rows per wave.
mkdir -p gpurun_out
timeout 300 python tools/op_list.py 64 > gpurun_out/r04_oplist_v2_fewer_slices.log 2>&1; tail -1 gpurun_out/r04_oplist_v2_fewer_slices.log
grep "layernorm" gpurun_out/r04_oplist_v2_fewer_slices.log | awk '{for(i=1;i<=NF;i++) if($i ~ /^ms=/) {split($i,a,"="); s+=a[2]} else if ($i=="ms=") s+=$(i+1)} END{print "layernorm total ms", s}'
awk '$2=="conv3x3" && $0 ~ /GF= +1[24][0-9]\.[0-9]|GF= +241\.6|GF= +147\.6/' gpurun_out/r04_oplist_v2_fewer_slices.log | tail -12
timeout 1500 python -m pytest tests/test_unet_ops.py -x -q -m gpu 2>&1 | tail -4
