"""LayerNorm at the UNet's shapes (64 images), plain tensors and stream pairs: ms per call and TB/s on the bytes each form moves (read 2 or 3 B, write 2 B).
(Round 5 used it for the A/B of a full-lane kernel behind MVE_LN_FLAT -- profiles/r05_ln_flat_ab.log; that kernel is not in the library.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib, ops  # noqa: E402
from mvedit_amd.ops import dt as _dt  # noqa: E402

dev = torch.device('cuda:0')
for (M, C) in [(262144, 320), (65536, 640), (16384, 1280), (32768, 320)]:
    g = torch.Generator().manual_seed(1)
    hi, lo = ops.split_pair(torch.randn(M, C, generator=g) * 2, torch.float16)
    hi, lo = hi.to(dev), lo.to(dev)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    out = torch.empty(M, C, dtype=torch.float16, device=dev)
    res = []
    for pair in (False, True):
        run = lambda: _lib.call('mve_layernorm_pair', _dt(hi), _lib.ptr(hi), C, _lib.ptr(out), C, M, C, _lib.ptr(gam), _lib.ptr(bet), 1e-5,
                                _lib.ptr(lo) if pair else None, _lib.stream_ptr(dev))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            run()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 30
        res.append((ms, M * C * (5 if pair else 4) / ms / 1e9))
    ref = torch.nn.functional.layer_norm(hi.float() + ops.lo8_to_float(lo.cpu()).to(dev), (C,), gam, bet, 1e-5)
    err = float((out.float() - ref).abs().max())
    print(f'M={M} C={C}: plain {res[0][0]:.4f} ms ({res[0][1]:.2f} TB/s)   pair {res[1][0]:.4f} ms ({res[1][1]:.2f} TB/s)   max |pair out - fp32 ref| {err:.2e}', flush=True)
