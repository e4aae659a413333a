"""Timing-only ablations of k_attention3's default configuration (k_attention3<..., ABL>, mve_attention_tune bits 8-19): which part of the
64-key tile loop the time goes to.  Results of ablated launches are WRONG by construction.  python tools/ab_attention_ablate.py"""
import sys

import torch

sys.path.insert(0, '.')
from mvedit_amd import _lib, ops

NAMES = {2048: 'baseline', 1: 'no s_barrier', 3: 'no barrier, no vmcnt wait', 7: 'no barrier / vmcnt / LDS-DMA', 8: 'no v_exp', 16: 'no row-max chain',
         24: 'no exp, no max', 32: 'no QK^T MFMA', 64: 'no PV MFMA', 96: 'no MFMA at all', 128: 'no K fragment reads', 256: 'no V^T fragment reads',
         384: 'no LDS fragment reads', 512: 'no permlane16 swaps', 536: 'no exp / max / swaps', 927: 'MFMAs (+ cvt) only', 480: 'softmax VALU + DMA + barrier only',
         1023: 'loop skeleton (cvt only)'}


def timed(fn, it=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


import ctypes
tune = _lib.raw('mve_attention_tune')
prof = _lib.raw('mve_attention_profile')
B, L, heads, d = 64, 4096, 8, 40
C = heads * d
torch.manual_seed(0)
qkv = torch.randn(B * L, 3 * C, device='cuda', dtype=torch.float16)
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
qp = (q.float() * (d ** -0.5 * 1.4426950408889634)).half()
tiles = B * heads * (L // 32) * (L // 64) / 1024.0           # 32-query x 64-key wave tiles per SIMD
for rnd in range(2):
    for abl in NAMES:
        tune(9 | (abl << 8))
        buf = (ctypes.c_ulonglong * 4)()
        prof(buf)
        ms = timed(lambda: ops.attention(qp, k, v, B, L, L, heads, d, prescaled=True))
        prof(buf)
        cyc, ticks, waves = buf[0], buf[1], max(buf[2], 1)
        ntile = L // 64
        print(f'round {rnd} ABL {abl:5d} {NAMES[abl]:40s} {ms:8.3f} ms  {4.0 * B * heads * L * L * d / ms / 1e9:7.0f} "TFLOP/s"  {ms * 1e6 / tiles:7.1f} ns/tile/SIMD'
              f' | per wave: {cyc / waves / ntile:7.0f} cyc/tile  {ticks * 10.0 / waves / ntile:7.0f} ns/tile  clock {cyc / max(ticks, 1) / 10.0:5.2f} GHz', flush=True)
tune(9)
