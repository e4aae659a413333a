#!/bin/bash
mkdir -p gpurun_out
python -m mvedit_amd.build > gpurun_out/build.log 2>&1 || tail -5 gpurun_out/build.log
timeout 1800 python -m pytest tests/test_unet_ops.py tests/test_pipeline_mixin.py -q -m gpu -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_gpu.log | head -40
for v in 4 8 16 32; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --views $v > gpurun_out/bench_v$v.log 2>&1; tail -1 gpurun_out/bench_v$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('views', d['config']['views'], d['ms_per_step'], d['roofline']['per_class_ms'], d['roofline']['per_class_tflops'])"
done
