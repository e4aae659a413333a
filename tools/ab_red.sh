#!/bin/bash
# tools/ab_red.py under a kernel trace: the host-paired times of that script are host-bound (a ctypes call costs more than these kernels); the trace
# gives GPU durations.  Output: gpurun_out/ab_red_rows.csv (tools/trace_rows.py) + the label log.
REPO=$PWD
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $REPO/gpurun_out/abred_trace -o t -- python $REPO/tools/ab_red.py ${1:-8} > $REPO/gpurun_out/ab_red_labels.log 2>&1
python $REPO/tools/trace_rows.py $REPO/gpurun_out/abred_trace > $REPO/gpurun_out/ab_red_rows.csv
rm -rf $REPO/gpurun_out/abred_trace
wc -l $REPO/gpurun_out/ab_red_rows.csv
