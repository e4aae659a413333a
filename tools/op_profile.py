"""Per-op timing of the benchmark plan (64 images): aggregates by label, prints the top entries."""
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine  # noqa: E402
from mvedit_amd import synthetic as U  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = UNet2DConditionEngine(SD15_CONFIG, torch.float16)
g = torch.Generator().manual_seed(0)
sd = {}
for name, shape in U.param_shapes(SD15_CONFIG).items():
    sd[name] = torch.randn(shape, generator=g, dtype=torch.float16) * 0.02
eng.load_state_dict(sd)
x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
for _ in range(2):
    eng(x, 499, ctx)
agg = defaultdict(lambda: [0.0, 0.0, 0])
tot = defaultdict(float)
for it in range(3):
    _, rows = eng.profile(x, 499, ctx)
    for cls, lab, fl, ms in rows:
        a = agg[(cls, lab)]
        a[0] += ms / 3; a[1] += fl / 3; a[2] += 1
        tot[cls] += ms / 3
print({k: round(v, 2) for k, v in tot.items()}, 'total', round(sum(tot.values()), 2))
for (cls, lab), (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f'{cls:10s} {lab:45s} n={n // 3:3d} ms={ms:7.3f} TF/s={fl / ms / 1e9 if ms else 0:7.1f}')
