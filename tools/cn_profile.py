"""Per-op timing of one SD-1.5 ControlNet over 64 images."""
import os, sys, torch
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import synthetic as SY
from mvedit_amd.controlnet import ControlNetEngine
from mvedit_amd.unet import SD15_CONFIG
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cn = ControlNetEngine.from_state_dict(SY.make_controlnet_state_dict(dict(SD15_CONFIG), dtype=torch.float16), dict(SD15_CONFIG), torch.float16)
x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
cond = torch.rand(B, 3, 512, 512, device='cuda', dtype=torch.float16)
dn, md = cn.new_outputs(B, 64, 64)
cn.run(x, 499, ctx, cond, 1.0, dn, md, False)
rows = cn.run(x, 499, ctx, cond, 1.0, dn, md, False, profile=True)
agg = defaultdict(lambda: [0.0, 0.0, 0])
for cls, lab, fl, ms in rows:
    a = agg[(cls, lab)]; a[0] += ms; a[1] += fl; a[2] += 1
print('total', round(sum(r[3] for r in rows), 2))
for (cls, lab), (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f'{cls:10s} {lab:40s} n={n:3d} ms={ms:7.3f} TF/s={fl / ms / 1e9 if ms else 0:7.1f}')
for cls, lab, fl, ms in rows[:22]:
    if 'cond' in lab or lab == 'silu': print(f'   {lab:32s} {ms:7.3f} ms')
