"""Writes tests/golden/recon_loss_ref.npz by EXECUTING the reference's own statements for the image-space part of one NeRF optimisation
iteration -- lib/pipelines/mvedit_3d_pipeline.py, method `nerf_optim`, from `out_rgbs = outputs['image']...` to `loss = loss +
entropy_loss` (:540-609) -- cut out of the file where it lies with `ast` and run on the CPU in float64 over seeded renderer outputs,
together with the in-tree functions those statements call: `depth_to_normal` (lib/core/utils/geometry_utils.py:119-148), `tv_loss` /
`TVLoss` (lib/models/losses/tv_loss.py), `l1_loss_mod` / `L1LossMod` (lib/models/losses/pixelwise_loss.py) and `Tonemapping`
(lib/models/decoders/tonemapping.py).  Absent third-party pieces are stood in for: mmgen's `weighted_loss` decorator and `L1Loss` base
(mmgen 0.7: element-wise loss * weight, then mean), the registry decorator (identity).  Gradients are torch autograd's.
Run from the repo root (needs /root/reference):  python tests/golden/make_recon_loss_golden.py"""
import ast
import functools
import importlib.util
import math
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import textwrap
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'recon_loss_ref.npz')


def weighted_loss(fn):
    """mmgen.models.losses.utils.weighted_loss (mmgen 0.7.x), the subset the reference uses: weight multiplies the element-wise loss,
    reduction 'mean' (avg_factor None) averages it."""
    @functools.wraps(fn)
    def wrapper(*args, weight=None, reduction='mean', avg_factor=None, **kwargs):
        loss = fn(*args, **kwargs)
        if weight is not None:
            loss = loss * weight
        assert avg_factor is None and reduction == 'mean'
        return loss.mean()
    return wrapper


class L1Loss(nn.Module):
    """mmgen.models.losses.L1Loss constructor (what L1LossMod inherits)."""

    def __init__(self, loss_weight=1.0, reduction='mean', loss_name='loss_l1'):
        super().__init__()
        self.loss_weight, self.reduction = loss_weight, reduction


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _exec_ref(path, names, ns):
    """exec the named top-level definitions of a reference file (its third-party imports are absent here)"""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), path, 'exec'), ns)
    return ns


def _loss_block():
    """the statements of nerf_optim's loop body from the first use of outputs['image'] to the entropy term"""
    path = 'lib/pipelines/mvedit_3d_pipeline.py'
    src = open(os.path.join(REF, path)).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == 'nerf_optim')
    loop = next(n for n in ast.walk(fn) if isinstance(n, ast.For) and getattr(n.target, 'id', '') == 'inverse_step_id')
    seg = [ast.get_source_segment(src, s) for s in loop.body]
    first = next(i for i, s in enumerate(seg) if s.startswith("out_rgbs = outputs['image']"))
    last = next(i for i, s in enumerate(seg) if s.startswith('loss = loss + entropy_loss'))
    body = [s for s in loop.body[first:last + 1]]
    return compile(ast.Module(body, []), path, 'exec'), (body[0].lineno, body[-1].end_lineno)


def make_case(seed, P, ps, tonemap, is_init, init_shaded, use_normal, use_depth, tm):
    g = torch.Generator().manual_seed(seed)
    dd = torch.float64
    N = P * ps * ps
    # renderer outputs of P patches of ps x ps rays (float64 leaves)
    # alpha in [0.05, 1] with a block of exact ones and two rows of exact zeros (patch 0): next to the zeros depth_to_normal differences
    # points at distance 1e6, where even float32 vs float64 disagree -- the tests compare those pixels only loosely
    alpha = 0.05 + 0.95 * torch.rand(N, generator=g, dtype=dd)
    alpha.view(P, ps, ps)[-1, -3:, -3:] = 1.0
    alpha.view(P, ps, ps)[0, :2] = 0.0
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, ps, dtype=dd), torch.linspace(-1, 1, ps, dtype=dd), indexing='ij')
    dirs = torch.stack([xx * 0.27, yy * 0.27, torch.ones_like(xx)], -1)[None].expand(P, -1, -1, -1).contiguous()
    dirs = dirs + torch.randn(P, 1, 1, 3, generator=g, dtype=dd) * 0.02
    inv_r = (0.25 + 0.08 * torch.rand(P, ps, ps, generator=g, dtype=dd)
             + 0.05 * torch.sin(3 * xx + torch.arange(P, dtype=dd)[:, None, None]))   # 1 / r of the surface
    depth = (inv_r.reshape(N) * alpha)
    image = torch.rand(N, 3, generator=g, dtype=dd) * alpha[:, None]
    cnt = torch.randint(2, 7, (N,), generator=g)
    M = int(cnt.sum())
    weights = torch.rand(M, generator=g, dtype=dd) * 0.4
    weights[::17] = 0.0
    ts = torch.stack([torch.rand(M, generator=g, dtype=dd) * 3 + 1, torch.rand(M, generator=g, dtype=dd) * 0.03 + 1e-3], -1)
    tgt = dict(target_rgbs=torch.rand(P, ps, ps, 3, generator=g, dtype=dd), target_m_blur=torch.rand(P, ps, ps, 1, generator=g, dtype=dd),
               target_dir=dirs, target_n=torch.rand(P, ps, ps, 3, generator=g, dtype=dd),
               target_depth=torch.rand(P, ps, ps, 1, generator=g, dtype=dd) * 0.4)
    cam_w = torch.rand(P, generator=g, dtype=dd) + 0.5
    lights = F.normalize(torch.randn(P, 3, generator=g, dtype=dd), dim=-1)
    leaves = dict(image=image, weights_sum=alpha, depth=depth, weights=weights)
    for v in leaves.values():
        v.requires_grad_(True)
    hyper = dict(is_init=is_init, init_shaded=init_shaded, use_normal=use_normal, use_depth=use_depth, ambient_light=0.2,
                 normal_reg_weight=2.5, depth_weight=0.7, entropy_weight=1.3, bg_width=0.015, cam_weights_mean=float(cam_w.mean()))
    ns = dict(torch=torch, F=F, math=math, debug=False, outputs=dict(leaves, ts=[ts]), **tgt, **hyper)
    ns['target_w'] = cam_w[:, None, None, None].expand(-1, ps, ps, 1)
    ns['target_lights'] = lights[:, None, None, :].expand(-1, ps, ps, 3)
    ns['normal_bg'] = torch.tensor([0.5, 0.5, 1.0], dtype=dd)
    ns['self'] = types.SimpleNamespace(nerf=types.SimpleNamespace(patch_size=ps, pixel_loss=ns_ref['L1LossMod'](loss_weight=1.2), bg_color=1.0),
                                       tonemapping=tm if tonemap else None)
    ns['loss_tv'] = ns_ref['TVLoss'](loss_weight=1.0, power=1.5)
    ns['depth_to_normal'] = ns_ref['depth_to_normal']
    code, lines = _loss_block()
    exec(code, ns)
    loss = ns['loss']
    grads = torch.autograd.grad(loss, list(leaves.values()))
    out = dict(P=P, ps=ps, tonemap=int(tonemap), cam_w=cam_w, lights=lights, ts=ts, cnt=cnt, **{k: v.detach() for k, v in leaves.items()},
               **tgt, **{k: float(v) for k, v in hyper.items()},
               loss=loss.detach(), pixel_rgb_loss=ns['pixel_rgb_loss'].detach(), alphas_loss=ns['alphas_loss'].detach(),
               normal_reg_loss=ns['normal_reg_loss'].detach(), entropy_loss=ns['entropy_loss'].detach(),
               depth_loss=ns['depth_loss'].detach() if use_depth else torch.zeros(()),
               out_rgbs=ns['out_rgbs'].detach().reshape(P, ps, ps, 3), out_normals=ns['out_normals'].detach(),
               out_normals_fg=ns['out_normals_fg'].detach(), out_normals_fg_weight=ns['out_normals_fg_weight'].detach(),
               **{'g_' + k: gv for k, gv in zip(leaves, grads)})
    return {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}, lines


ns_ref = {}


def main():
    base = dict(torch=torch, nn=nn, F=F, MODULES=_Registry(), weighted_loss=weighted_loss, L1Loss=L1Loss)
    ns_ref.update(base)
    _exec_ref('lib/core/utils/geometry_utils.py', {'depth_to_normal'}, ns_ref)
    _exec_ref('lib/models/losses/tv_loss.py', {'tv_loss', 'TVLoss'}, ns_ref)
    _exec_ref('lib/models/losses/pixelwise_loss.py', {'l1_loss_mod', 'L1LossMod'}, ns_ref)
    spec = importlib.util.spec_from_file_location('ref_tonemapping', os.path.join(REF, 'lib/models/decoders/tonemapping.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tm = mod.Tonemapping().double()
    out = {}
    cases = [dict(seed=1, P=3, ps=8, tonemap=True, is_init=False, init_shaded=False, use_normal=True, use_depth=True),
             dict(seed=2, P=2, ps=12, tonemap=False, is_init=False, init_shaded=False, use_normal=False, use_depth=False),
             dict(seed=3, P=2, ps=8, tonemap=True, is_init=True, init_shaded=False, use_normal=True, use_depth=False),
             dict(seed=4, P=2, ps=8, tonemap=True, is_init=True, init_shaded=True, use_normal=False, use_depth=True)]
    for i, c in enumerate(cases):
        res, lines = make_case(tm=tm, **c)
        out.update({f'c{i}_{k}': v for k, v in res.items()})
        print(i, c, 'loss', float(res['loss']), 'block lines', lines)
    out['n_cases'] = np.asarray(len(cases))
    out['lut_x'], out['lut_y'] = tm.lut_x.numpy(), tm.lut_y.numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
