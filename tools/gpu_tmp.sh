#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_segmentor.py -m gpu -x -q -s -p no:cacheprovider 2>&1 | tail -40 | cut -c1-400 | tee gpurun_out/r03_segmentor_tests.log
