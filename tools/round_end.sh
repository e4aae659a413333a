#!/bin/bash
# Round-end measurement with the final binary: the default bench command, the 1-rank RCCL smoke of the multi-GPU path, the strong-scaling preview.
# (tools/profile_round.sh: the rocprofv3 passes over the same bench command; tools/gpu_suite.sh: the whole GPU test suite.)
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_final.log 2>&1; tail -c 600 gpurun_out/bench_final.log; echo
MVE_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-extra --no-secondary --no-cpu-baseline > gpurun_out/bench_dist_smoke.log 2>&1; tail -c 400 gpurun_out/bench_dist_smoke.log; echo
python tools/scale_preview.py > gpurun_out/scale_preview.log 2>&1; grep -v amdgpu gpurun_out/scale_preview.log
# (round 5) everything of a round's end in one gpurun call: PROF_TAG=rNN bash tools/profile_round.sh; python __graft_entry__.py smoke; bash tools/gpu_suite.sh; bash tools/round_end.sh
