#!/bin/bash
# Upsample2D as phase convs: one launch vs four launches vs the 3x3 form, at 64 and 8 images; op + engine tests first.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_unet_ops.py -m gpu -x -q -k "upsample_conv_phases" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_vae.py tests/test_unet.py -m gpu -x -q -k "vae or tiny or small_all or sd15_256px or graph_replay or decode or chunking or topology" 2>&1 | tail -4
for B in 64 8; do
  python tools/op_list.py $B > gpurun_out/oplist_b${B}_one.log 2>&1; echo "B=$B one launch : $(tail -1 gpurun_out/oplist_b${B}_one.log)"; grep upsample gpurun_out/oplist_b${B}_one.log
  MVE_PHASES_ONE_LAUNCH=0 python tools/op_list.py $B > gpurun_out/oplist_b${B}_four.log 2>&1; echo "B=$B four       : $(tail -1 gpurun_out/oplist_b${B}_four.log)"; grep upsample gpurun_out/oplist_b${B}_four.log
  MVE_UPSAMPLE_PHASES=0 python tools/op_list.py $B > gpurun_out/oplist_b${B}_3x3.log 2>&1; echo "B=$B 3x3        : $(tail -1 gpurun_out/oplist_b${B}_3x3.log)"; grep upsample gpurun_out/oplist_b${B}_3x3.log
done
