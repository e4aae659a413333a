#!/bin/bash
mkdir -p gpurun_out
python -m mvedit_amd.build > gpurun_out/build.log 2>&1 || tail -5 gpurun_out/build.log
timeout 1800 python -m pytest tests/test_dmtet.py -q -m gpu -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_gpu.log | head -40
