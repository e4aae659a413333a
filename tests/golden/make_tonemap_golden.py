"""Writes tests/golden/tonemap_ref.npz by EXECUTING the reference's Tonemapping class (lib/models/decoders/tonemapping.py, imported
from /root/reference where it lies: it only needs torch).  The shading case wraps the reference's own lut / inverse_lut around the
inline arithmetic of lib/pipelines/mvedit_3d_pipeline.py:1372-1384 (restated in oracle/tonemap_oracle.py).
Run from the repo root (needs /root/reference):  python tests/golden/make_tonemap_golden.py"""
import importlib.util
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import tonemap_oracle as T  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tonemap_ref.npz')


def main():
    spec = importlib.util.spec_from_file_location('ref_tonemapping', '/root/reference/lib/models/decoders/tonemapping.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tm = mod.Tonemapping()
    g = torch.Generator().manual_seed(0)
    out = dict(lut_x=tm.lut_x.numpy(), lut_y=tm.lut_y.numpy())
    xlog = torch.cat([torch.rand(4000, generator=g) * 16 - 11, tm.lut_x, torch.tensor([-20.0, 7.5])])       # knots and both outsides
    xlin = torch.cat([torch.rand(4000, generator=g) * 4, torch.tensor([0.0, 1e-7, 1.0, 8.0])])
    yv = torch.cat([torch.rand(4000, generator=g) * 1.4 - 0.2, tm.lut_y])
    out.update(x_log=xlog.numpy(), x_lin=xlin.numpy(), y=yv.numpy(),
               lut_log=tm.lut(xlog).numpy(), lut_lin=tm.lut(xlin, input_mode='linear').numpy(),
               inv_log=tm.inverse_lut(yv).numpy(), inv_lin=tm.inverse_lut(yv, output_mode='linear').numpy())
    # gradients of the four maps as torch autograd gives them for the reference's expressions (float64 leaves, away from the knots)
    tm64 = mod.Tonemapping().double()
    for key, src, fn in (('log', xlog[:4000], lambda v: tm64.lut(v)), ('lin', xlin, lambda v: tm64.lut(v, input_mode='linear')),
                         ('inv_log', yv[:4000], lambda v: tm64.inverse_lut(v)), ('inv_lin', yv[:4000], lambda v: tm64.inverse_lut(v, output_mode='linear'))):
        leaf = src.double().requires_grad_(True)
        gr, = torch.autograd.grad(fn(leaf).sum(), leaf)
        out['grad_' + key] = gr.numpy()
    # a rendered batch: 3 views of 20 x 24 pixels, alpha in [0, 1] with exact zeros and ones
    b, S1, S2 = 3, 20, 24
    alpha = torch.rand(1, b, S1, S2, 1, generator=g)
    alpha[0, 0, :4] = 0.0
    alpha[0, 1, :4] = 1.0
    rgba = torch.cat([torch.rand(1, b, S1, S2, 3, generator=g) * alpha, alpha], dim=-1)
    normal_fg = torch.rand(1, b, S1, S2, 3, generator=g)
    lights = torch.nn.functional.normalize(torch.randn(b, 3, generator=g), dim=-1)
    out.update(rgba=rgba.numpy(), normal_fg=normal_fg.numpy(), cam_lights=lights.numpy(),
               shaded_tm=T.shade_views(rgba, normal_fg, lights, 0.1, 1.0, lut_fn=tm.lut, inv_fn=tm.inverse_lut).numpy(),
               shaded_plain=T.shade_views(rgba, normal_fg, lights, 0.1, 1.0).numpy())
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
