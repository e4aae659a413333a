#!/bin/bash
# kernel trace of the small-batch forward (tools/fwd_small.py B) -> per-kernel durations + gaps (tools/trace_gaps.py)
REPO=$PWD
B=${1:-8}
TAG=${2:-b$B}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $REPO/gpurun_out/trace_$TAG -o t -- python $REPO/tools/fwd_small.py $B 4 > $REPO/gpurun_out/trace_$TAG.log 2>&1
tail -2 $REPO/gpurun_out/trace_$TAG.log
python $REPO/tools/trace_gaps.py $REPO/gpurun_out/trace_$TAG --list > $REPO/gpurun_out/trace_gaps_$TAG.txt 2>&1
head -40 $REPO/gpurun_out/trace_gaps_$TAG.txt
rm -rf $REPO/gpurun_out/trace_$TAG
