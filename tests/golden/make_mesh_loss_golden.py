"""Writes tests/golden/mesh_loss_ref.npz by EXECUTING the reference's own statements for the image-space part of one mesh optimisation
iteration -- lib/pipelines/mvedit_3d_pipeline.py, method `mesh_optim`, from `out_alphas = render_out['rgba']...` to the
`loss = loss + alphas_loss + normal_reg_loss + ...` line -- cut out of the file where it lies with `ast` and run on the CPU in float64 over
seeded renderer outputs, with the in-tree `depth_to_normal`, `tv_loss` / `TVLoss`, `l1_loss_mod` / `L1LossMod` they call (same stand-ins
for the absent mmgen pieces as make_recon_loss_golden.py).  The two mesh regularisers inside the same `if` block are pinned separately
(make_mesh_reg_golden.py) and return 0 here.  Gradients are torch autograd's.
Run from the repo root (needs /root/reference):  python tests/golden/make_mesh_loss_golden.py"""
import ast
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_recon_loss_golden import REF, L1Loss, _Registry, _exec_ref, weighted_loss  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mesh_loss_ref.npz')


def _loss_block():
    path = 'lib/pipelines/mvedit_3d_pipeline.py'
    src = open(os.path.join(REF, path)).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == 'mesh_optim')
    loop = next(n for n in ast.walk(fn) if isinstance(n, ast.For) and getattr(n.target, 'id', '') == 'inverse_step_id')
    seg = [ast.get_source_segment(src, s) for s in loop.body]
    first = next(i for i, s in enumerate(seg) if s.startswith("out_alphas = render_out['rgba']"))
    last = next(i for i, s in enumerate(seg) if s.startswith('if not mesh_is_simplified:') and 'normal_reg_loss' in s)
    body = loop.body[first:last + 1]
    return compile(ast.Module(body, []), path, 'exec'), (body[0].lineno, body[-1].end_lineno)


def make_case(ref, seed, n, S, use_normal, simplified):
    g = torch.Generator().manual_seed(seed)
    dd = torch.float64
    N = n * S * S
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S, dtype=dd), torch.linspace(-1, 1, S, dtype=dd), indexing='ij')
    dirs = torch.stack([xx * 0.27, yy * 0.27, torch.ones_like(xx)], -1)[None].expand(n, -1, -1, -1).contiguous()
    alpha = (torch.rand(n, S, S, 1, generator=g, dtype=dd) * 1.3 - 0.1).clamp(0, 1)           # exact zeros and ones, values below 1e-3
    alpha[0, 0, :3] = 5e-4
    rgba = torch.cat([torch.rand(n, S, S, 3, generator=g, dtype=dd) * alpha, alpha], -1)[None]
    nbg = torch.tensor([0.5, 0.5, 1.0], dtype=dd)
    normal = (torch.rand(n, S, S, 3, generator=g, dtype=dd) * alpha + nbg * (1 - alpha))[None]
    depth = (0.3 + 0.08 * torch.rand(n, S, S, generator=g, dtype=dd) + 0.05 * torch.sin(3 * xx))[None]   # 1 / z of the rendered surface
    leaves = dict(rgba=rgba.requires_grad_(True), normal=normal.requires_grad_(True))
    cam_w = torch.rand(n, generator=g, dtype=dd) + 0.5
    ns = dict(torch=torch, F=F, debug=False, render_out=dict(leaves, depth=depth), depth_to_normal=ref['depth_to_normal'],
              target_rgbs=torch.rand(n, S, S, 3, generator=g, dtype=dd), target_m_blur=torch.rand(n, S, S, 1, generator=g, dtype=dd),
              target_m_erode=(torch.rand(n, S, S, 1, generator=g, dtype=dd) * 1.4 - 0.2).clamp(0, 1), target_dir=dirs,
              target_n=torch.rand(n, S, S, 3, generator=g, dtype=dd), use_normal=use_normal, normal_bg=nbg,
              target_w=cam_w[:, None, None, None].expand(-1, S, S, 1), cam_weights_mean=float(cam_w.mean()), normal_reg_weight=2.5,
              mesh_is_simplified=simplified, mesh_normal_reg_weight=0.0, loss_tv=ref['TVLoss'](loss_weight=1.0, power=1.5),
              laplacian_smooth_loss=lambda v, f: torch.zeros((), dtype=dd), normal_consistency=lambda fn_, f: torch.zeros((), dtype=dd),
              in_mesh=types.SimpleNamespace(v=None, f=None, face_normals=None))
    ns['self'] = types.SimpleNamespace(nerf=types.SimpleNamespace(pixel_loss=ref['L1LossMod'](loss_weight=1.2)))
    code, lines = _loss_block()
    exec(code, ns)
    loss = ns['loss']
    ext = [torch.randn(n, S, S, 3, generator=g, dtype=dd) * 1e-3 for _ in range(2)]           # stand-ins for the patch-loss gradients
    total = loss * 0.6 + (ns['out_rgbs'] * ext[0]).sum() + (ns['out_normals'] * ext[1]).sum()
    g_plain = torch.autograd.grad(loss, list(leaves.values()), retain_graph=True, allow_unused=True)
    g_plain = [torch.zeros_like(l) if gg is None else gg for gg, l in zip(g_plain, leaves.values())]
    g_ext = torch.autograd.grad(total, list(leaves.values()))
    out = dict(n=n, S=S, use_normal=int(use_normal), simplified=int(simplified), cam_w=cam_w, depth=depth[0], rgba=rgba.detach()[0],
               normal=normal.detach()[0], **{k: ns[k] for k in ('target_rgbs', 'target_m_blur', 'target_m_erode', 'target_dir', 'target_n')},
               cam_weights_mean=ns['cam_weights_mean'], normal_reg_weight=ns['normal_reg_weight'], loss=loss.detach(),
               pixel_rgb_loss=ns['pixel_rgb_loss'].detach(), alphas_loss=torch.zeros(()) if simplified else ns['alphas_loss'].detach(),
               normal_reg_loss=torch.zeros(()) if simplified else ns['normal_reg_loss'].detach(), out_rgbs=ns['out_rgbs'].detach(),
               out_normals=ns['out_normals'].detach(), out_normals_cos=ns['out_normals_cos'].detach(), ext_rgb=ext[0], ext_nrm=ext[1],
               g_rgba=g_plain[0][0], g_normal=g_plain[1][0], gx_rgba=g_ext[0][0], gx_normal=g_ext[1][0])
    return {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}, lines


def main():
    ref = dict(torch=torch, nn=nn, F=F, MODULES=_Registry(), weighted_loss=weighted_loss, L1Loss=L1Loss)
    _exec_ref('lib/core/utils/geometry_utils.py', {'depth_to_normal'}, ref)
    _exec_ref('lib/models/losses/tv_loss.py', {'tv_loss', 'TVLoss'}, ref)
    _exec_ref('lib/models/losses/pixelwise_loss.py', {'l1_loss_mod', 'L1LossMod'}, ref)
    out = {}
    cases = [dict(seed=1, n=2, S=10, use_normal=True, simplified=False), dict(seed=2, n=3, S=8, use_normal=False, simplified=False),
             dict(seed=3, n=2, S=8, use_normal=True, simplified=True)]
    for i, c in enumerate(cases):
        res, lines = make_case(ref, **c)
        out.update({f'c{i}_{k}': v for k, v in res.items()})
        print(i, c, 'loss', float(res['loss']), 'block lines', lines)
    out['n_cases'] = np.asarray(len(cases))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
