#!/bin/bash
# Round 4: the whole GPU test suite.
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^$" > gpurun_out/r04_pytest_gpu_full.log
grep -n "passed\|failed\|FAILED\|ERROR" gpurun_out/r04_pytest_gpu_full.log | tail -30
