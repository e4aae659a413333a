#!/bin/bash
timeout 300 python tools/tracer_profile.py bfloat16 8 2>&1 | tail -1
timeout 300 python tools/tracer_profile.py bfloat16 32 2>&1 | tail -1
timeout 600 python -m pytest tests/test_segmentor.py -x -q -m gpu -k "chunks or oracle" 2>&1 | tail -2
