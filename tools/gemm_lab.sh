#!/bin/bash
REPO=$PWD
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
MVE_LIB_TAG=lab timeout 600 rocprofv3 --kernel-trace -d $REPO/gpurun_out/lab_trace -o t -- python $REPO/tools/ab_gemm_lab.py > $REPO/gpurun_out/gemm_lab_labels.log 2>&1
python $REPO/tools/trace_rows.py $REPO/gpurun_out/lab_trace > $REPO/gpurun_out/gemm_lab_rows.csv
rm -rf $REPO/gpurun_out/lab_trace
wc -l $REPO/gpurun_out/gemm_lab_rows.csv
