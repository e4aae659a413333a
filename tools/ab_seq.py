"""A/B: real split-K + reducer vs sequential-slice emulation in the big-tile kernel (level-2 shapes at 64 images)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, _lib
from tools.microbench import timeit
tune = _lib.raw('mve_gemm_tune')
dt, dev, B = torch.float16, 'cuda', 64
for (H, C1, Cout) in [(16, 1280, 1280), (16, 2560, 1280), (16, 640, 1280)]:
    x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
    w = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt) * (9 * C1) ** -0.5
    r = []
    for flag in (1 << 29, 0):
        tune(256 | flag)
        f = lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=True)
        o = f()[0]; r.append((timeit(f, 2, 6) * 1e3, o))
    print(f'conv H={H} {C1}->{Cout}: split+reducer {r[0][0]:.3f} ms | seq {r[1][0]:.3f} ms | equal={torch.equal(r[0][1], r[1][1])}', flush=True)
for (M, N, K, rpi) in [(B * 256, 1280, 5120, 256), (B * 256, 1280, 1280, 256), (B * 256, 3840, 1280, 256)]:
    a = torch.randn(M, K, device=dev, dtype=dt); w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    r = []
    for flag in (1 << 29, 0):
        tune(256 | flag)
        f = lambda: ops.gemm(a, w, rows_per_image=rpi)
        o = f(); r.append((timeit(f, 2, 6) * 1e3, o))
    print(f'gemm M={M} N={N} K={K}: split+reducer {r[0][0]:.3f} ms | seq {r[1][0]:.3f} ms | equal={torch.equal(r[0][1], r[1][1])}', flush=True)
