#!/bin/bash
# One GPU visit: parity tests (full log), microbench.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export MVE_DUMP_REF_GOLDEN=$PWD/gpurun_out/raymarching_ref_gfx950.npz
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
python -c "import torch;print(torch.cuda.get_device_name(0), torch.version.hip)" >> gpurun_out/device.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
tail -45 gpurun_out/microbench.log
