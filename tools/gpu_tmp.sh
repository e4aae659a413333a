#!/bin/bash
timeout 900 python -m pytest tests/test_unet.py tests/test_controlnet.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py 2>gpurun_out/r03_bench_final2.err > gpurun_out/r03_bench_final2.log
tail -1 gpurun_out/r03_bench_final2.log | cut -c1-300
