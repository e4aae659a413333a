#!/bin/bash
# round 5, GPU call 1: the new default mode (residual pair + phase convs), canonical pair arithmetic on the 128-row kernel, rows-of-the-timed-batch parity,
# multi-stream small-batch probe, clock-limiter probe
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_unet.py tests/test_pipeline_mixin.py tests/test_controlnet.py -q -m gpu -s -x 2>&1 | grep -v "^$" > gpurun_out/r05_pytest_unet_family.log
tail -n 15 gpurun_out/r05_pytest_unet_family.log
timeout 400 python -m pytest tests/test_unet_ops.py -q -m gpu -x -k "pair or phase or invarian or bit_identical" 2>&1 | tail -n 5 > gpurun_out/r05_pytest_ops_pair.log; cat gpurun_out/r05_pytest_ops_pair.log
timeout 600 python bench.py > gpurun_out/r05_bench_v0.log 2>&1; tail -c 1500 gpurun_out/r05_bench_v0.log; echo
timeout 300 python tools/multistream_probe.py 8 16 2 > gpurun_out/r05_multistream_probe.log 2>&1; grep -v amdgpu gpurun_out/r05_multistream_probe.log
MVE_RESIDUAL_PAIR=0 timeout 300 python tools/multistream_probe.py 8 16 > gpurun_out/r05_multistream_probe_plain.log 2>&1; grep -v amdgpu gpurun_out/r05_multistream_probe_plain.log
timeout 200 python tools/limiter_probe.py > gpurun_out/r05_limiter.log 2>&1; grep -A12 "==== summary" gpurun_out/r05_limiter.log
timeout 200 python tools/scale_preview.py > gpurun_out/r05_scale_preview_v0.log 2>&1; grep -v amdgpu gpurun_out/r05_scale_preview_v0.log
