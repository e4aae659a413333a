#!/bin/bash
mkdir -p gpurun_out
python -m mvedit_amd.build > gpurun_out/build.log 2>&1 || tail -5 gpurun_out/build.log
export MVE_GEMM_VARIANT=2
timeout 1200 python -m pytest tests/test_unet_ops.py tests/test_unet.py tests/test_pipeline_mixin.py -q -m gpu -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_v2.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_v2.log | head -40
timeout 600 python tools/microbench.py > gpurun_out/microbench_v2.log 2>&1
grep -E "gemm|conv" gpurun_out/microbench_v2.log | cut -c1-150
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_rs.log 2>&1; tail -1 gpurun_out/bench_rs.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms'], d['roofline']['per_class_tflops'])"
