#!/bin/bash
timeout 900 python -m pytest tests/test_triplane.py -x -q -m gpu > gpurun_out/r03_pytest_triplane.log 2>&1
tail -30 gpurun_out/r03_pytest_triplane.log
