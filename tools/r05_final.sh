#!/bin/bash
# round 5, end of round: rocprofv3 passes over the bench command, the whole GPU suite, the default bench command, the 1-rank RCCL smoke, smoke(), previews
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
PROF_TAG=r05 bash tools/profile_round.sh > gpurun_out/r05_profile_round.log 2>&1; tail -n 5 gpurun_out/r05_profile_round.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 2
bash tools/gpu_suite.sh 2>&1 | tail -n 12
bash tools/round_end.sh 2>&1 | tail -n 12
timeout 200 python tools/op_list.py 8 > gpurun_out/r05_oplist_8images.log 2>&1; tail -n 1 gpurun_out/r05_oplist_8images.log
