#!/bin/bash
timeout 600 python tools/ab_tracer_ops.py > gpurun_out/r03_ab_tracer_ops_v4.log 2>&1
tail -16 gpurun_out/r03_ab_tracer_ops_v4.log
timeout 300 python tools/tracer_profile.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_segmentor.py -x -q -m gpu 2>&1 | tail -2
