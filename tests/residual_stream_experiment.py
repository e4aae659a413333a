"""Where the end-to-end 16-bit error of the UNet comes from: the oracle with storage rounding at every op boundary
(PyTorch's half path) against the same with the RESIDUAL STREAM kept in fp32 (x + f(x) of ResnetBlock2D / BasicTransformerBlock /
Transformer2DModel not rounded), both measured against pure fp32 arithmetic over the same 16-bit weights, at the benchmark latent size.
CPU only (torch); ~1 minute."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))          # tests/ may use the oracle (tools/ may not)
from oracle import unet_oracle as UO
from mvedit_amd.unet import SD15_CONFIG

cfg = SD15_CONFIG
for dt in (torch.float16, torch.bfloat16):
    sd = {k: v.to(dt).float() for k, v in UO.make_state_dict(cfg, seed=1234).items()}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 4, 64, 64, generator=g).to(dt).float()
    ctx = torch.randn(1, 77, 768, generator=g).to(dt).float()
    t = torch.tensor([499.0])
    with torch.no_grad():
        ref = UO.unet_forward(sd, cfg, x, t, ctx)
        q = UO.quantizer(dt)
        emu = UO.unet_forward(sd, cfg, x, t, ctx, q=q)
        # residual stream in fp32: a context whose residual quantizer is the identity
        c = UO._Ctx(sd, cfg, q, None)
        c.qr = lambda v: v
        emb, res, h, c = UO.unet_enc(sd, cfg, x, t, ctx, 1, q, None, _c=c)
        out, _ = UO.unet_dec(sd, cfg, emb, res, h, ctx, 1, None, None, q, None, _c=c)
    rel = lambda a: float((a - ref).norm() / ref.norm())
    print(f'{str(dt):16s} rel-L2 vs fp32: every op output rounded {rel(emu):.3e}   residual stream in fp32 {rel(out):.3e}')
