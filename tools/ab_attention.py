"""Same-box A/B of the attention kernel variants (mve_attention_tune) on the UNet's self-attention shapes; prints time, TFLOP/s and the
difference between the variants' outputs.  python tools/ab_attention.py"""
import sys

import torch

sys.path.insert(0, '.')
from mvedit_amd import _lib, ops


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


tune = _lib.raw('mve_attention_tune')
for B, L, heads, d in ((64, 4096, 8, 40), (8, 4096, 8, 40), (64, 1024, 8, 80), (64, 256, 8, 160), (16, 4096, 5, 64), (32, 8192, 8, 40)):
    C = heads * d
    qkv = torch.randn(B * L, 3 * C, device='cuda', dtype=torch.float16)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    outs = []
    variants = {40: (0, 8, 9, 10, 11, 12), 64: (0, 4)}.get(d, (0, 6))
    for variant in variants:
        tune(variant)
        ms = timed(lambda: ops.attention(q, k, v, B, L, L, heads, d))
        outs.append(ops.attention(q, k, v, B, L, L, heads, d))
        fl = 4.0 * B * heads * L * L * d
        print(f'B={B:3d} L={L:5d} d={d:3d} variant {variant}: {ms:8.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s')
    if d == 40:                                     # the executors' call: Q carries softmax_scale * log2(e)
        qp = (q.float() * (d ** -0.5 * 1.4426950408889634)).half()
        for variant in (8, 9, 10, 11, 12):
            tune(variant)
            ms = timed(lambda: ops.attention(qp, k, v, B, L, L, heads, d, prescaled=True))
            print(f'B={B:3d} L={L:5d} d={d:3d} variant {variant} prescaled: {ms:8.3f} ms  {4.0 * B * heads * L * L * d / ms / 1e9:7.0f} TFLOP/s')
    tune(11)
    print('      max |variant - default| =', [(v, (o.float() - outs[0].float()).abs().max().item()) for v, o in zip(variants[1:], outs[1:])])
