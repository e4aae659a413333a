"""Same-box A/B of the K-sliced launches one rank of an 8-GPU job runs (B images, default 8): in-kernel slice fold (mve_gemm_red_tune(1)) against
partials + k_splitk_reduce (0).  Times are per call over back-to-back launches (HIP events)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
red = _lib.raw('mve_gemm_red_tune')
deep = _lib.raw('mve_gemm_deep_tune')
OFF = 1 << 30                                     # mve_gemm_deep_tune bit 30: row-panel-major block order everywhere (rounds 1-5)
MODES = [('r05', 0, OFF), ('wmajor', 0, 0), ('fold128', 1, 0), ('foldpp', 2, 0), ('foldboth', 3, 0)]      # (label, in-kernel fold, four-stage ring up to N blocks | OFF)
dt = torch.float16


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device='cuda') * scale).to(dt)


print(f'{"op":46s} ' + ' '.join(f'{m[0]:>9s}' for m in MODES) + '   (us per call)')
for (hw, N, K, what) in [(1024, 640, 2560, 'L1 ff.out'), (256, 1280, 1280, 'L2 proj/to_out'), (256, 1280, 5120, 'L2 ff.out'), (64, 1280, 1280, 'L3 proj/to_out'), (64, 1280, 5120, 'L3 ff.out')]:
    M = B * hw
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias, res = torch.randn(N, device='cuda'), rnd(M, N)
    t = []
    for (_, r_, d_) in MODES:
        red(r_); deep(d_)
        t.append(timeit(lambda: ops.gemm(a, w, bias=bias, residual=res, rows_per_image=hw)))
    print(f'linear {what:18s} M={M:5d} N={N:4d} K={K:5d}   ' + ' '.join(f'{x:9.1f}' for x in t))
for (H, C1, C2, Cout, what) in [(32, 320, 0, 640, 'L1 conv1'), (32, 640, 0, 640, 'L1 conv'), (16, 640, 0, 1280, 'L2 conv1'), (16, 1280, 0, 1280, 'L2 conv'), (16, 1280, 1280, 1280, 'L2 up conv1'),
                              (8, 1280, 0, 1280, 'L3 conv'), (8, 1280, 1280, 1280, 'L3 up conv1')]:
    x1 = rnd(B * H * H, C1)
    x2 = rnd(B * H * H, C2) if C2 else None
    wt = rnd(Cout, C1 + C2, 3, 3, scale=(9 * (C1 + C2)) ** -0.5)
    w_k, wflag = ops.pack_conv_weight(wt, True)
    bias = torch.randn(Cout, device='cuda')
    t = []
    for (_, r_, d_) in MODES:
        red(r_); deep(d_)
        t.append(timeit(lambda: ops.conv3x3(x1, w_k, B, H, H, x2=x2, bias=bias, flags=wflag)))
    print(f'conv   {what:18s} {H:2d}x{H:<2d} C={C1 + C2:4d}->{Cout:4d}        ' + ' '.join(f'{x:9.1f}' for x in t))
# un-sliced small launches (the 128-row kernel without K slices)
for (hw, N, K, what) in [(1024, 640, 640, 'L1 proj/to_out'), (256, 3840, 1280, 'L2 qkv'), (64, 3840, 1280, 'L3 qkv'), (64, 10240, 1280, 'L3 geglu'), (1, 1280, 1280, 'time_emb.linear_2')]:
    M = B * hw
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias = torch.randn(N, device='cuda')
    t = []
    for (_, r_, d_) in MODES:
        red(r_); deep(d_)
        t.append(timeit(lambda: ops.gemm(a, w, bias=bias, rows_per_image=hw)))
    print(f'linear {what:18s} M={M:5d} N={N:4d} K={K:5d}   ' + ' '.join(f'{x:9.1f}' for x in t))
red(-1); deep(-1)
