#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_unet_ops.py -m gpu -x -q -k "gemm or geglu or identical or pingpong or splitk" -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/ab_gelu.py 2>&1 | grep -v "no-gelu\|amdgpu" | tee gpurun_out/r03_ab_gelu_v2.log
