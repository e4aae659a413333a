"""Linear-class shapes of the 64-image forward: the 256-row ping-pong tile (one block of 8 waves per CU) against the 128 x 160 kernel (two
independent 4-wave blocks per CU: a block's epilogue overlaps the other block's K loop).  python tools/ab_linear_small.py [images]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, _lib  # noqa: E402
from tools.microbench import timeit  # noqa: E402

tune = _lib.raw('mve_gemm_tune')
dt, dev = torch.float16, 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tot = [0.0, 0.0]
for (M, N, K, rpi, cnt, fl, res) in [(B * 4096, 960, 320, 4096, 5, 0, 0), (B * 4096, 320, 320, 4096, 10, 0, 0), (B * 4096, 320, 320, 4096, 15, 0, 1), (B * 4096, 2560, 320, 4096, 5, 1, 0), (B * 4096, 320, 1280, 4096, 5, 0, 1),
                                     (B * 1024, 1920, 640, 1024, 5, 0, 0), (B * 1024, 640, 640, 1024, 10, 0, 0), (B * 1024, 640, 640, 1024, 15, 0, 1), (B * 1024, 5120, 640, 1024, 5, 1, 0), (B * 1024, 640, 2560, 1024, 5, 0, 1),
                                     (B * 256, 3840, 1280, 256, 5, 0, 0), (B * 256, 1280, 1280, 256, 8, 0, 0), (B * 256, 1280, 1280, 256, 12, 0, 1), (B * 256, 10240, 1280, 256, 5, 1, 0), (B * 256, 1280, 5120, 256, 5, 0, 1),
                                     (B * 64, 1280, 1280, 64, 4, 0, 1), (B * 64, 10240, 1280, 64, 1, 1, 0), (B * 64, 1280, 5120, 64, 1, 0, 1)]:
    a = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    bias = torch.randn(N, device=dev, dtype=torch.float32)
    r = torch.randn(M, N, device=dev, dtype=dt) if res else None
    outs, ts = [], []
    for small in (0, 1):
        tune(0 if small else 1)
        f = lambda: ops.gemm(a, w, bias=bias, residual=r, flags=ops.GEGLU if fl else 0, rows_per_image=rpi)
        outs.append(f())
        ts.append(timeit(f, 2, 5) * 1e3)
        tot[small] += ts[-1] * cnt
    flops = 2 * M * N * K
    print(f'gemm  M={M:7d} N={N:5d} K={K:5d} x{cnt:2d} geglu={fl} res={res}  pp256 {ts[0]:7.3f} ms {flops / ts[0] / 1e9:6.0f} TF | small128x160 {ts[1]:7.3f} ms {flops / ts[1] / 1e9:6.0f} TF'
          f' | equal={torch.equal(outs[0], outs[1])}', flush=True)
tune(1)
print(f'linear total per forward-set: pp256 {tot[0]:.2f} ms, small {tot[1]:.2f} ms', flush=True)
