"""Writes tests/golden/shading_fun_ref.npz by EXECUTING the reference's own shading-function factories -- `MVEdit3DPipeline.make_shading_fun`,
`make_nerf_shading_fun`, `make_nerf_albedo_shading_fun` (lib/pipelines/mvedit_3d_pipeline.py:410-450, cut out of the class with `ast`) -- over
the reference's Tonemapping class (float64) and a stand-in decoder, with torch autograd for the gradients w.r.t. albedo / decoder colour and
world_normal.  Run from the repo root (needs /root/reference):  python tests/golden/make_shading_fun_golden.py"""
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import ast
import importlib.util
import os
import types

import numpy as np
import torch

REF = '/root/reference/lib/pipelines/mvedit_3d_pipeline.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shading_fun_ref.npz')


def main():
    cls = next(n for n in ast.parse(open(REF).read()).body if isinstance(n, ast.ClassDef) and n.name == 'MVEdit3DPipeline')
    ns = dict(torch=torch)
    for fn in cls.body:
        if isinstance(fn, ast.FunctionDef) and fn.name.startswith('make_') and fn.name.endswith('shading_fun'):
            exec(compile(ast.Module([fn], []), REF, 'exec'), ns)
    spec = importlib.util.spec_from_file_location('ref_tonemapping', '/root/reference/lib/models/decoders/tonemapping.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tm = mod.Tonemapping().double()
    g = torch.Generator().manual_seed(0)
    b, h, w = 2, 12, 10
    fg = torch.rand(1, b, h, w, generator=g) > 0.4
    N = int(fg.sum())
    lights = torch.nn.functional.normalize(torch.randn(b, h, w, 3, generator=g, dtype=torch.float64), dim=-1)
    albedo = (torch.rand(N, 3, generator=g, dtype=torch.float64) * 0.9 + 0.02).requires_grad_(True)
    normal = torch.nn.functional.normalize(torch.randn(N, 3, generator=g, dtype=torch.float64), dim=-1).requires_grad_(True)
    pos = torch.rand(N, 3, generator=g, dtype=torch.float64) - 0.5
    W = torch.rand(3, 3, generator=g, dtype=torch.float64)
    W.requires_grad_(True)
    point_albedo = lambda x: torch.sigmoid(x @ W)                                   # the stand-in decoder colour
    gy = torch.randn(N, 3, generator=g, dtype=torch.float64)
    out = dict(fg=fg.numpy(), lights=lights.numpy(), albedo=albedo.detach().numpy(), normal=normal.detach().numpy(), pos=pos.numpy(), W=W.detach().numpy(),
               gy=gy.numpy(), lut_x=tm.lut_x.numpy(), lut_y=tm.lut_y.numpy())
    for tag, tmo in (('tm', tm), ('plain', None)):
        me = types.SimpleNamespace(tonemapping=tmo, nerf=types.SimpleNamespace(decoder=types.SimpleNamespace(
            point_decode=lambda xyzs, dirs, code: (None, point_albedo(xyzs[0])[None]))))
        f1 = ns['make_shading_fun'](me, lights, 0.2)
        y = f1(world_pos=pos, albedo=albedo, world_normal=normal, fg_mask=fg)
        ga, gn = torch.autograd.grad((y * gy).sum(), (albedo, normal))
        out.update({f'{tag}_mesh_out': y.detach().numpy(), f'{tag}_mesh_g_albedo': ga.numpy(), f'{tag}_mesh_g_normal': gn.numpy()})
        f2 = ns['make_nerf_shading_fun'](me, None, lights, 0.2)
        y = f2(world_pos=pos, albedo=albedo, world_normal=normal, fg_mask=fg)
        gw, gn = torch.autograd.grad((y * gy).sum(), (W, normal))
        out.update({f'{tag}_nerf_out': y.detach().numpy(), f'{tag}_nerf_g_W': gw.numpy(), f'{tag}_nerf_g_normal': gn.numpy()})
    f3 = ns['make_nerf_albedo_shading_fun'](me, None)
    out['albedo_fun_out'] = f3(world_pos=pos, albedo=albedo).detach().numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'N =', N)


if __name__ == '__main__':
    main()
