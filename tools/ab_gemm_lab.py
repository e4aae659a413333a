"""Timing-only ablation of the 128-row GEMM / conv kernel on small launches (lab build: MVE_BUILD_TAG=lab MVE_BUILD_DEFS=MVE_GEMM_LAB python -m
mvedit_amd.build; run with MVE_LIB_TAG=lab).  MVE_GEMM_LAB_BITS: 16 no MFMAs, 32 no LDS-DMA after the first tiles, 64 no epilogue.  Results are garbage."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib, ops  # noqa: E402

B = 8
dt = torch.float16
_lib.raw('mve_gemm_red_tune')(0)
deep = _lib.raw('mve_gemm_deep_tune')


def timeit(fn, n=10):
    """Host-paired time is useless here (a ctypes call + two allocations cost more than the kernel): the launches are counted by
    tools/trace_rows.py from a rocprofv3 kernel trace of this script -- 13 launches of the main kernel per (op, ring, case), the last 10 averaged."""
    for _ in range(3 + n):
        fn()
    torch.cuda.synchronize()
    return 0.0


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device='cuda') * scale).to(dt)


CASES = [('full', 0), ('no mfma', 16), ('no dma', 32), ('no epi', 64), ('dma only', 16 | 64), ('mfma only', 32 | 64), ('shell', 16 | 32 | 64)]
print(f'{"op":44s} ring ' + ' '.join(f'{c[0]:>9s}' for c in CASES))
ops_ = []
for (hw, N, K, what, rpi) in [(64, 1280, 1280, 'L3 proj', 64), (64, 1280, 5120, 'L3 ff.out', 64), (256, 1280, 1280, 'L2 proj', 256), (1, 1280, 1280, 'time_emb2', 0), (64, 1280, 1280, 'L3 proj unsliced', 0)]:
    M = B * hw
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias = torch.randn(N, device='cuda')
    ops_.append((f'linear {what:16s} M={M:5d} N={N} K={K}', lambda a=a, w=w, bias=bias, rpi=rpi: ops.gemm(a, w, bias=bias, rows_per_image=rpi)))
for (H, C1, Cout, what) in [(8, 1280, 1280, 'L3 conv'), (16, 1280, 1280, 'L2 conv')]:
    x1 = rnd(B * H * H, C1)
    wt = rnd(Cout, C1, 3, 3, scale=(9 * C1) ** -0.5)
    w_k, wflag = ops.pack_conv_weight(wt, True)
    bias = torch.randn(Cout, device='cuda')
    ops_.append((f'conv   {what:16s} {H}x{H} C={C1}->{Cout}', lambda x1=x1, w_k=w_k, bias=bias, H=H, wflag=wflag: ops.conv3x3(x1, w_k, B, H, H, bias=bias, flags=wflag)))
for name, fn in ops_:
    for ring in (0, 256):
        deep(ring)
        t = []
        for _, bits in CASES:
            os.environ['MVE_GEMM_LAB_BITS'] = str(bits)
            t.append(timeit(fn))
        print(f'{name:44s} {"4" if ring else "2":>4s}', flush=True)
