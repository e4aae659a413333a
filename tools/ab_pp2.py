"""Linear-class shapes: the 256 x 320 ping-pong tile (one block per CU) against the 256 x 160 three-slot tile (two blocks per CU,
mve_gemm_tune bit 26) + bitwise equality.  python tools/ab_pp2.py [images]"""
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
SHAPES = [(B * 4096, 960, 320, 4096, 5, 0, 0), (B * 4096, 320, 320, 4096, 10, 0, 0), (B * 4096, 320, 320, 4096, 15, 0, 1), (B * 4096, 2560, 320, 4096, 5, 1, 0), (B * 4096, 320, 1280, 4096, 5, 0, 1),
          (B * 1024, 1920, 640, 1024, 5, 0, 0), (B * 1024, 640, 640, 1024, 10, 0, 0), (B * 1024, 640, 640, 1024, 15, 0, 1), (B * 1024, 5120, 640, 1024, 5, 1, 0), (B * 1024, 640, 2560, 1024, 5, 0, 1),
          (B * 256, 3840, 1280, 256, 5, 0, 0), (B * 256, 1280, 1280, 256, 8, 0, 0), (B * 256, 1280, 1280, 256, 12, 0, 1), (B * 256, 10240, 1280, 256, 5, 1, 0), (B * 256, 1280, 5120, 256, 5, 0, 1)]
if len(sys.argv) > 2:          # quick correctness pass on small problems first (a hang or a fault shows up here under a short timeout)
    SHAPES = [(512, 320, 64, 0, 1, 0, 0), (512, 320, 320, 0, 1, 0, 1), (1024, 640, 640, 0, 1, 1, 0), (768, 160, 128, 0, 1, 0, 1), (2048, 960, 1280, 0, 1, 0, 0)]
tot = [0.0, 0.0]
for (M, N, K, rpi, cnt, fl, res) in SHAPES:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    bias = torch.randn(N, device=dev, dtype=torch.float32)
    r = torch.randn(M, N, device=dev, dtype=dt) if res else None
    outs, ts = [], []
    for two in (0, 1):
        tune(1 | (1 << 26) if two else 1)
        f = lambda: ops.gemm(a, w, bias=bias, residual=r, flags=ops.GEGLU if fl else 0, rows_per_image=rpi)
        outs.append(f())
        torch.cuda.synchronize()
        ts.append(timeit(f, 2, 6) * 1e3)
        tot[two] += ts[-1] * cnt
    flops = 2 * M * N * K
    nbad = int((outs[0] != outs[1]).sum())
    print(f'gemm  M={M:7d} N={N:5d} K={K:5d} x{cnt:2d} geglu={fl} res={res}  320-wide {ts[0]:7.3f} ms {flops / ts[0] / 1e9:6.0f} TF | 2x160 {ts[1]:7.3f} ms {flops / ts[1] / 1e9:6.0f} TF'
          f' | equal={torch.equal(outs[0], outs[1])} nbad={nbad} finite={bool(torch.isfinite(outs[1].float()).all())}', flush=True)
tune(256)
print(f'linear total per forward-set: 320-wide {tot[0]:.2f} ms, 2 x 160 {tot[1]:.2f} ms', flush=True)
