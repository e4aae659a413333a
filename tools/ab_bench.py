"""Same-box A/B of the whole 64-image forward under different mve_gemm_tune settings (box-to-box variance is ~10 %)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine
from mvedit_amd import synthetic as U
from tools.microbench import timeit
tune = _lib.raw('mve_gemm_tune')
utune = _lib.raw('mve_unet_tune')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
sd = {n: torch.randn(sh, generator=g, dtype=torch.float16) * 0.02 for n, sh in U.param_shapes(SD15_CONFIG).items()}
engs = {}
for name, v in (('fused-shortcut', 1), ('separate-shortcut', 0)):
    utune(v)
    engs[name] = UNet2DConditionEngine(SD15_CONFIG, torch.float16)
    engs[name].load_state_dict(sd)
utune(1)
x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
outs = {}
for rep in range(3):
    for name, eng in engs.items():
        outs[name] = eng(x, 499, ctx)[0]
        t = timeit(lambda: eng(x, 499, ctx), 1, 3) * 1e3
        print(f'{name:18s} {t:8.3f} ms', flush=True)
a, b = outs['fused-shortcut'].float(), outs['separate-shortcut'].float()
print('max |diff| / max |out| =', ((a - b).abs().max() / b.abs().max()).item())
