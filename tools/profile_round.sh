#!/bin/bash
# rocprofv3 passes over the bench command.  Summaries land in gpurun_out/prof_*; copy what matters to profiles/.
REPO=$PWD
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-extra"
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_stats -o bench -- $BENCH > $REPO/gpurun_out/prof_stats.log 2>&1
tail -3 $REPO/gpurun_out/prof_stats.log
BENCH1="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-extra --no-op-timing"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $REPO/gpurun_out/prof_fetch -o bench -- $BENCH1 > $REPO/gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $REPO/gpurun_out/prof_write -o bench -- $BENCH1 > $REPO/gpurun_out/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 -d $REPO/gpurun_out/prof_mfma -o bench -- $BENCH1 > $REPO/gpurun_out/prof_mfma.log 2>&1
cd $REPO/gpurun_out && find . -name "*.csv" | head -30 && du -sh .
# keep only per-kernel aggregates of the PMC passes (raw per-dispatch csv can be large)
python $REPO/tools/summarize_prof.py $REPO/gpurun_out > $REPO/gpurun_out/prof_summary.txt 2>&1
tail -60 $REPO/gpurun_out/prof_summary.txt
# pmc_traffic.json from the same databases, then drop the raw rocpd databases: gpurun merges at most 64 MiB back (the MFMA pass alone is > 100 MB)
python $REPO/tools/make_pmc_traffic.py $REPO/gpurun_out profiles/${PROF_TAG:-rNN}_rocprof_summary.txt > $REPO/gpurun_out/pmc_traffic.json 2>$REPO/gpurun_out/pmc_traffic.err
rm -rf $REPO/gpurun_out/prof_stats $REPO/gpurun_out/prof_fetch $REPO/gpurun_out/prof_write $REPO/gpurun_out/prof_mfma
# afterwards, in the repo: cp gpurun_out/prof_summary.txt profiles/rNN_rocprof_summary.txt; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
