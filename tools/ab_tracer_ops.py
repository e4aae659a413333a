"""Per-shape timing of the TRACER-B7 encoder's three operators (expand 1x1, depthwise + squeeze sums, project 1x1 with the SE gate) over the
distinct MBConv shapes of EfficientNet-B7 at 640^2, 8 images: microseconds, achieved GB/s of algorithmic traffic, share of the chunk."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvedit_amd import synthetic as SY
from mvedit_amd.segmentor import TracerUniversalB7Engine, ACT_SWISH

dev = torch.device('cuda:0')
B = 8
seg = TracerUniversalB7Engine(input_image_size=640, batch_size=B, torch_dtype='bfloat16', erosion=1, device=dev).load_state_dict(SY.make_tracer_state_dict(3))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


H = 320
seen = {}
tot = dict(expand=0.0, dw=0.0, project=0.0)
floor = dict(expand=0.0, dw=0.0, project=0.0)
for n, (k, st, e, cin, cout, se, pad) in enumerate(seg.blocks):
    blk = seg.p[f'encoder._blocks.{n}']
    mid = cin * e
    Ho = (H + pad[0] + pad[1] - k) // st + 1
    key = (k, st, e, cin, cout, H)
    if key not in seen:
        x = torch.randn(B * H * H, cin, device=dev).to(torch.bfloat16)
        xm = torch.randn(B * H * H, mid, device=dev).to(torch.bfloat16)
        xo = torch.randn(B * Ho * Ho, mid, device=dev).to(torch.bfloat16)
        gate = torch.rand(B, mid, device=dev)
        r = {}
        if e != 1:
            r['expand'] = (timeit(lambda: seg._pw(x, B, H * H, blk['expand'], act=ACT_SWISH)), B * H * H * (cin + mid) * 2)
        r['dw'] = (timeit(lambda: seg._dwconv_pool(xm, B, H, H, mid, blk['dw'], k, st, pad)), B * (H * H + Ho * Ho) * mid * 2)
        res = torch.randn(B * Ho * Ho, cout, device=dev).to(torch.bfloat16) if (st == 1 and cin == cout) else None
        r['project'] = (timeit(lambda: seg._pw(xo, B, Ho * Ho, blk['project'], gate=gate, residual=res)), B * Ho * Ho * (mid + cout * (2 if res is not None else 1)) * 2)
        seen[key] = r
        print(f'block {n:2d} k{k} s{st} cin {cin:4d} mid {mid:4d} cout {cout:4d} {H:3d}->{Ho:3d}: ' +
              '  '.join(f'{nm} {us:7.1f} us {by / us / 1e3:6.0f} GB/s' for nm, (us, by) in r.items()), flush=True)
    for nm, (us, by) in seen[key].items():
        tot[nm] += us
        floor[nm] += by / 4.6e6
    H = Ho
print('per chunk (ms):', {k: round(v / 1e3, 2) for k, v in tot.items()}, ' at 4.6 TB/s:', {k: round(v / 1e3, 2) for k, v in floor.items()})
