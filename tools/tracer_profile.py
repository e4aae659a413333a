"""Time the TRACER-B7 engine on one 8-view chunk at 640^2 (the shape of adapter3d_mixin.py:14-19) -- run under rocprofv3 --kernel-trace --stats
for the per-kernel table:  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tracer -- python tools/tracer_profile.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvedit_amd import synthetic as SY
from mvedit_amd.segmentor import TracerUniversalB7Engine

dev = torch.device('cuda:0')
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bfloat16'
seg = TracerUniversalB7Engine(input_image_size=640, batch_size=8, torch_dtype=dtype, erosion=1, device=dev).load_state_dict(SY.make_tracer_state_dict(3))
g = torch.Generator(device='cpu').manual_seed(5)
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 8
x = torch.rand(NV, 3, 512, 512, generator=g).to(dev)
for _ in range(2):
    m = seg(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    m = seg(x)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / n
print(f'tracer_b7 {NV} views 512^2 -> 640^2 {dtype}: {t * 1e3:.2f} ms per call = {t * 1e3 / NV:.2f} ms per view; mask mean {float(m.mean()):.4f}')
