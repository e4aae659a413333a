"""A/B of the two main loops of the 256-row GEMM/conv tile (two-stage: gemm_big.hip, ping-pong: gemm_pp.hip) on the SD-1.5 shapes of the
benchmark (64 images) + bitwise equality.  Columns: two-stage ("big"), ping-pong ("pp")."""
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
MODES = {0: 1 | (1 << 27), 1: 1}
tot = {0: 0.0, 1: 0.0}
rows = []
for (H, C1, Cout, cnt) in [(64, 320, 320, 7), (64, 640, 320, 2), (64, 960, 320, 1), (32, 320, 640, 1), (32, 640, 640, 6), (32, 1280, 640, 1),
                           (32, 1920, 640, 1), (32, 960, 640, 1), (16, 640, 1280, 1), (16, 1280, 1280, 8), (16, 2560, 1280, 2),
                           (16, 1920, 1280, 1), (8, 1280, 1280, 8), (8, 2560, 1280, 3)]:
    x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
    w = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt) * (9 * C1) ** -0.5
    outs, ts = [], []
    for big in (0, 1):
        tune(MODES[big])
        f = lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=True)
        outs.append(f()[0])
        ts.append(timeit(f, 2, 5) * 1e3)
        tot[big] += ts[-1] * cnt
    fl = 2 * B * H * H * Cout * 9 * C1
    rows.append(f'conv  H={H:3d} {C1:5d}->{Cout:5d} x{cnt}  big {ts[0]:7.3f} ms {fl / ts[0] / 1e9:6.0f} TF | pp {ts[1]:7.3f} ms {fl / ts[1] / 1e9:6.0f} TF'
                f' | equal={torch.equal(outs[0], outs[1])}')
    print(rows[-1], flush=True)
print(f'conv total per forward-set: big {tot[0]:.2f} ms, pp {tot[1]:.2f} ms', flush=True)
tot = {0: 0.0, 1: 0.0}
for (M, N, K, rpi, cnt, fl) in [(B * 4096, 960, 320, 4096, 5, 0), (B * 4096, 320, 320, 4096, 20, 0), (B * 4096, 2560, 320, 4096, 5, 1), (B * 4096, 320, 1280, 4096, 5, 0),
                                (B * 1024, 1920, 640, 1024, 5, 0), (B * 1024, 640, 640, 1024, 20, 0), (B * 1024, 5120, 640, 1024, 5, 1), (B * 1024, 640, 2560, 1024, 5, 0),
                                (B * 256, 3840, 1280, 256, 5, 0), (B * 256, 1280, 1280, 256, 20, 0), (B * 256, 10240, 1280, 256, 5, 1), (B * 256, 1280, 5120, 256, 5, 0),
                                (B * 64, 3840, 1280, 64, 1, 0), (B * 64, 1280, 1280, 64, 4, 0), (B * 64, 10240, 1280, 64, 1, 1), (B * 64, 1280, 5120, 64, 1, 0)]:
    a = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    outs, ts = [], []
    for big in (0, 1):
        tune(MODES[big])
        f = lambda: ops.gemm(a, w, flags=ops.GEGLU if fl else 0, rows_per_image=rpi)
        outs.append(f())
        ts.append(timeit(f, 2, 5) * 1e3)
        tot[big] += ts[-1] * cnt
    flops = 2 * M * N * K
    print(f'gemm  M={M:7d} N={N:5d} K={K:5d} x{cnt}  big {ts[0]:7.3f} ms {flops / ts[0] / 1e9:6.0f} TF | pp {ts[1]:7.3f} ms {flops / ts[1] / 1e9:6.0f} TF'
          f' | equal={torch.equal(outs[0], outs[1])}', flush=True)
print(f'linear total per forward-set: big {tot[0]:.2f} ms, pp {tot[1]:.2f} ms', flush=True)
