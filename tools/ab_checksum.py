"""SHA-256 of the engine's outputs on fixed seeded inputs (UNet at the benchmark latent size, the VAE decoder, a few raw GEMM epilogue cases): two
builds whose arithmetic is meant to be identical (A/B library variants, see mvedit_amd/build.py MVE_BUILD_TAG) must print the same lines."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, synthetic as U  # noqa: E402
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine  # noqa: E402

sha = lambda t: hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
g = torch.Generator().manual_seed(0)
for dt in (torch.float16, torch.bfloat16):
    for (M, N, K, rv, res) in [(4096, 320, 320, 0, 0), (4096, 640, 960, 0, 1), (8192, 320, 2880, 1, 0), (8192, 1280, 640, 1, 1), (4000, 320, 328, 0, 1)]:
        a = torch.randn(M, K, generator=g).to('cuda', dt)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to('cuda', dt)
        b = torch.randn(N, generator=g).to('cuda')
        kw = {}
        if res:
            kw['residual'] = torch.randn(M, N, generator=g).to('cuda', dt)
        if rv:
            kw['rowvec'] = torch.randn(M // 4096 if M % 4096 == 0 else 1, N, generator=g).to('cuda')
            kw['rows_per_vec'] = 4096
        try:
            print(dt, M, N, K, rv, res, sha(ops.gemm(a, w, b, **kw)))
        except TypeError as e:
            print('gemm kwargs', e)
            break
eng = UNet2DConditionEngine.from_state_dict(U.make_state_dict(dict(SD15_CONFIG), seed=1234, dtype=torch.float16), dict(SD15_CONFIG), torch.float16, 'cuda')
x = torch.randn(8, 4, 64, 64, generator=g).to('cuda', torch.float16)
ctx = torch.randn(8, 77, 768, generator=g).to('cuda', torch.float16)
print('unet B=8', sha(eng(x, 499, ctx)[0]))
