#!/bin/bash
# One GPU visit: op tests under both GEMM variants, full tests, bench, microbench.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
MVE_GEMM_VARIANT=0 timeout 900 python -m pytest tests/test_unet_ops.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/pytest_ops_v0.log
tail -3 gpurun_out/pytest_ops_v0.log
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -100 > gpurun_out/pytest_gpu.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_gpu.log | head -40
timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
grep -E "gemm|conv|attention" gpurun_out/microbench.log | cut -c1-160
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -2 gpurun_out/bench.log
MVE_GEMM_VARIANT=0 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_v0.log 2>&1; tail -1 gpurun_out/bench_v0.log | cut -c1-400
