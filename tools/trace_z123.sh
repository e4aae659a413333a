#!/bin/bash
REPO=$PWD
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $REPO/gpurun_out/trace_z -o t -- python $REPO/tools/fwd_z123.py > $REPO/gpurun_out/trace_z123.log 2>&1
grep "step ms" $REPO/gpurun_out/trace_z123.log
python $REPO/tools/trace_rows.py $REPO/gpurun_out/trace_z > $REPO/gpurun_out/trace_z123_rows.csv
rm -rf $REPO/gpurun_out/trace_z
