#!/bin/bash
# Round 4, call 2: epilogue A/B per shape (bitwise column) + per-op list under each epilogue with the un-split chain default.
mkdir -p gpurun_out
timeout 600 python tools/ab_epilogue.py 64 > gpurun_out/r04_ab_epilogue_v1.log 2>&1; echo "ab rc=$?"
for m in 0 1 2; do MVE_GEMM_EPI_DIRECT=$m timeout 300 python tools/op_list.py 64 > gpurun_out/r04_oplist_epi$m.log 2>&1; echo "oplist epi=$m rc=$?"; tail -1 gpurun_out/r04_oplist_epi$m.log; done
grep -v "^/opt" gpurun_out/r04_ab_epilogue_v1.log | cut -c1-220
