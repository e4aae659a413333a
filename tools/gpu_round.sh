#!/bin/bash
# One GPU visit: parity tests (full log), smoke, bench, microbench.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
python -c "import torch;print(torch.cuda.get_device_name(0), torch.version.hip)" > gpurun_out/device.txt 2>&1
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; tail -5 gpurun_out/bench.log
