#!/bin/bash
mkdir -p gpurun_out
python -m mvedit_amd.build > gpurun_out/build.log 2>&1 || tail -5 gpurun_out/build.log
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -100 > gpurun_out/pytest_gpu.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_gpu.log | head -40
timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
grep -E "attention" gpurun_out/microbench.log | cut -c1-160
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms'], d['roofline']['per_class_tflops'])"
