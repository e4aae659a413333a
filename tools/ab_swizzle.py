"""Same-box A/B of the ping-pong tile's LDS ring swizzle: round 2's (row >> 2) & 3 (every ds_read_b128 fragment read 2-way bank
conflicted under the hardware's real lane groups) against (row >> 1) & 3 (conflict-free), and of the two-blocks-per-CU 256 x 160 tile,
on the conv and linear shapes of the 64-image forward.  Interleaved rounds, median.  python tools/ab_swizzle.py [images]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, _lib  # noqa: E402
from tools.microbench import timeit  # noqa: E402

tune = _lib.raw('mve_gemm_tune')
dt, dev = torch.float16, 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MODES = [('old swizzle', 1 | (1 << 25)), ('new swizzle', 1), ('2 x 160', 1 | (1 << 26))]
tot = [0.0, 0.0, 0.0]


def ab(name, f, flops, cnt, modes):
    ts = [[] for _ in modes]
    outs = []
    for i, (_, m) in enumerate(modes):
        tune(m)
        outs.append(f())
    for rnd in range(3):
        for i, (_, m) in enumerate(modes):
            tune(m)
            ts[i].append(timeit(f, 1, 4) * 1e3)
    med = [statistics.median(t) for t in ts]
    eq = all(torch.equal(outs[0], o) for o in outs[1:])
    print(name + ' x%2d ' % cnt + ' | '.join(f'{modes[i][0]} {med[i]:7.3f} ms {flops / med[i] / 1e9:6.0f} TF' for i in range(len(modes))) + f' | equal={eq}', flush=True)
    return med


ctot = [0.0, 0.0]
for (H, C1, Cout, cnt) in [(64, 320, 320, 7), (64, 640, 320, 2), (64, 960, 320, 1), (32, 320, 640, 1), (32, 640, 640, 6), (32, 1280, 640, 1),
                           (32, 1920, 640, 1), (32, 960, 640, 1), (16, 640, 1280, 1), (16, 1280, 1280, 8), (16, 2560, 1280, 2),
                           (16, 1920, 1280, 1), (8, 1280, 1280, 8), (8, 2560, 1280, 3)]:
    x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
    w = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt) * (9 * C1) ** -0.5
    f = lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=True)[0]
    med = ab(f'conv  H={H:3d} {C1:5d}->{Cout:5d}', f, 2 * B * H * H * Cout * 9 * C1, cnt, MODES[:2])
    for i in range(2):
        ctot[i] += med[i] * cnt
print(f'conv total per forward-set: old {ctot[0]:.2f} ms, new {ctot[1]:.2f} ms', flush=True)
for (M, N, K, rpi, cnt, fl, res) in [(B * 4096, 960, 320, 4096, 5, 0, 0), (B * 4096, 320, 320, 4096, 10, 0, 0), (B * 4096, 320, 320, 4096, 15, 0, 1), (B * 4096, 2560, 320, 4096, 5, 1, 0), (B * 4096, 320, 1280, 4096, 5, 0, 1),
                                     (B * 1024, 1920, 640, 1024, 5, 0, 0), (B * 1024, 640, 640, 1024, 10, 0, 0), (B * 1024, 640, 640, 1024, 15, 0, 1), (B * 1024, 5120, 640, 1024, 5, 1, 0), (B * 1024, 640, 2560, 1024, 5, 0, 1),
                                     (B * 256, 3840, 1280, 256, 5, 0, 0), (B * 256, 1280, 1280, 256, 8, 0, 0), (B * 256, 1280, 1280, 256, 12, 0, 1), (B * 256, 10240, 1280, 256, 5, 1, 0), (B * 256, 1280, 5120, 256, 5, 0, 1)]:
    a = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    bias = torch.randn(N, device=dev, dtype=torch.float32)
    r = torch.randn(M, N, device=dev, dtype=dt) if res else None
    f = lambda: ops.gemm(a, w, bias=bias, residual=r, flags=ops.GEGLU if fl else 0, rows_per_image=rpi)
    med = ab(f'gemm  M={M:7d} N={N:5d} K={K:5d} geglu={fl} res={res}', f, 2 * M * N * K, cnt, MODES)
    for i in range(3):
        tot[i] += med[i] * cnt
tune(256)
print(f'linear total per forward-set: old swizzle {tot[0]:.2f} ms, new swizzle {tot[1]:.2f} ms, 2 x 160 {tot[2]:.2f} ms', flush=True)
