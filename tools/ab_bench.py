"""Same-box A/B of the whole 64-image forward under different mve_gemm_tune settings (box-to-box variance is ~10 %)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine
from mvedit_amd import synthetic as U
from tools.microbench import timeit
tune = _lib.raw('mve_gemm_tune')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = UNet2DConditionEngine(SD15_CONFIG, torch.float16)
g = torch.Generator().manual_seed(0)
eng.load_state_dict({n: torch.randn(sh, generator=g, dtype=torch.float16) * 0.02 for n, sh in U.param_shapes(SD15_CONFIG).items()})
x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
configs = {'default(256)': 256, 'no-seq': 256 | (1 << 29), 'big>=128': 128, 'big>=512': 512}
outs = {}
for rep in range(2):
    for name, v in configs.items():
        tune(v)
        outs[name] = eng(x, 499, ctx)[0]
        t = timeit(lambda: eng(x, 499, ctx), 1, 3) * 1e3
        print(f'{name:14s} {t:8.3f} ms', flush=True)
print('bitwise equal across configs:', all(torch.equal(outs['default(256)'], o) for o in outs.values()))
