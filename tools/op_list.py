"""Every op of the benchmark plan (B images, default 64) with its HIP-event time, in execution order (mean of 3 profiled forwards)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine  # noqa: E402
from mvedit_amd import synthetic as U  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = UNet2DConditionEngine.from_state_dict(U.make_state_dict(dict(SD15_CONFIG), seed=1234, dtype=torch.float16), dict(SD15_CONFIG), torch.float16, 'cuda')
x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
for _ in range(2):
    eng(x, 499, ctx)
acc = None
for it in range(3):
    _, rows = eng.profile(x, 499, ctx)
    acc = [list(r) for r in rows] if acc is None else [[a[0], a[1], a[2], a[3] + r[3]] for a, r in zip(acc, rows)]
tot = {}
for i, (cls, lab, fl, ms) in enumerate(acc):
    ms /= 3
    tot[cls] = tot.get(cls, 0.0) + ms
    print(f'{i:4d} {cls:10s} {lab:42s} GF={fl / 1e9:9.1f} ms={ms:7.3f} TF/s={fl / ms / 1e9 if ms else 0:7.1f}')
print({k: round(v, 2) for k, v in tot.items()}, 'total', round(sum(tot.values()), 2))
