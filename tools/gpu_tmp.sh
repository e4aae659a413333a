#!/bin/bash
timeout 300 ./tools/probe/gather_probe > gpurun_out/r03_gather_probe.log 2>&1
cat gpurun_out/r03_gather_probe.log
timeout 900 python -m pytest tests/test_unet.py -x -q -m gpu 2>&1 | tail -2
