"""Pair launches (residual pair in, pair out) of the level-0 / level-1 GEMM and conv shapes at 64 images next to the same launches on plain tensors:
ms per launch and the effective TB/s on the bytes each form moves.  MVE_LIB_TAG selects an A/B build (mvedit_amd/build.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops  # noqa: E402

dev = 'cuda'
g = torch.Generator().manual_seed(2)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print('library tag:', os.environ.get('MVE_LIB_TAG', '(default)'))
for (M, N, K) in [(262144, 320, 320), (262144, 320, 1280), (65536, 640, 640), (65536, 640, 2560), (16384, 1280, 1280)]:
    a = (torch.randn(M, K, generator=g)).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    rh, rl = ops.split_pair(torch.randn(M, N, generator=g) * 2, torch.float16)
    rh, rl = rh.to(dev), rl.to(dev)
    t_plain = timeit(lambda: ops.gemm(a, w, bias=bias, residual=rh))
    t_pair = timeit(lambda: ops.gemm(a, w, bias=bias, residual=rh, residual_lo=rl, pair_out=True))
    by_plain, by_pair = M * K * 2 + M * N * 4, M * K * 2 + M * N * 6
    print(f'GEMM M={M} N={N} K={K}: plain {t_plain:.3f} ms ({by_plain / t_plain / 1e9:.2f} TB/s, {2 * M * N * K / t_plain / 1e9:.0f} TF/s)   '
          f'pair {t_pair:.3f} ms ({by_pair / t_pair / 1e9:.2f} TB/s)', flush=True)
for (B, H, C) in [(64, 64, 320), (64, 32, 640)]:
    x = torch.randn(B * H * H, C, generator=g).half().to(dev)
    w, fl = ops.pack_conv_weight((torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5).half())
    w = w.to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    rh, rl = ops.split_pair(torch.randn(B * H * H, C, generator=g) * 2, torch.float16)
    rh, rl = rh.to(dev), rl.to(dev)
    t_plain = timeit(lambda: ops.conv3x3(x, w, B, H, H, bias=bias, residual=rh, flags=fl, splitk=False))
    t_pair = timeit(lambda: ops.conv3x3(x, w, B, H, H, bias=bias, residual=rh, residual_lo=rl, flags=fl, splitk=False, pair_out=True))
    print(f'conv {B} x {H}x{H} x {C}->{C}: plain {t_plain:.3f} ms   pair {t_pair:.3f} ms', flush=True)
