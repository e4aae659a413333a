#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_unet.py tests/test_pipeline_mixin.py tests/test_controlnet.py -q -m gpu -s 2>&1 | grep -v "^$" > gpurun_out/r05_pytest_unet_family_v2.log
grep -E "rel-L2|l2=|passed|failed|FAILED|Error" gpurun_out/r05_pytest_unet_family_v2.log | cut -c1-250 | tail -n 60
timeout 600 python bench.py > gpurun_out/r05_bench_v1.log 2>&1; tail -c 600 gpurun_out/r05_bench_v1.log; echo
timeout 200 python tools/scale_preview.py > gpurun_out/r05_scale_preview_v1.log 2>&1; grep -v amdgpu gpurun_out/r05_scale_preview_v1.log
