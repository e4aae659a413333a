#!/bin/bash
# Upsample2D as four phase convs: op tests, engine tests, then timing against MVE_UPSAMPLE_PHASES=0 on the same box.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_unet_ops.py -m gpu -x -q -k "upsample_conv_phases or conv3x3" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_unet.py tests/test_image_enhancer.py -m gpu -x -q 2>&1 | tail -8
python tools/op_list.py 64 > gpurun_out/oplist_phases.log 2>&1; tail -1 gpurun_out/oplist_phases.log; grep upsample gpurun_out/oplist_phases.log
MVE_UPSAMPLE_PHASES=0 python tools/op_list.py 64 > gpurun_out/oplist_3x3.log 2>&1; tail -1 gpurun_out/oplist_3x3.log; grep upsample gpurun_out/oplist_3x3.log
python tools/vae_check.py 8 --detail > gpurun_out/vae_phases.log 2>&1; grep -E "^decode|upsample" gpurun_out/vae_phases.log
MVE_UPSAMPLE_PHASES=0 python tools/vae_check.py 8 > gpurun_out/vae_3x3.log 2>&1; grep -E "^decode|upsample" gpurun_out/vae_3x3.log
