#!/bin/bash
# The whole GPU test suite on a gpurun box (log under gpurun_out/; copy to profiles/rNN_pytest_gpu_*.log).
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^$" > gpurun_out/pytest_gpu_full.log
grep -n "passed\|failed\|FAILED\|ERROR" gpurun_out/pytest_gpu_full.log | tail -30
