"""GPU debugging aid for tests/test_mesh_loss.py::test_mesh_optim_iteration_on_native_kernels_only: the same loop with (1) a finite-difference
check of d loss / d vertices along smooth directions (uniform scaling, a translation, a random low-frequency field) through the whole
native chain auto_normal -> MeshRenderer.forward -> mesh_optim_loss, term by term, and (2) a longer loss history.
python tools/debug_mesh_loop.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mvedit_amd.mesh_ops import Mesh, MeshRenderer, mesh_regularizers  # noqa: E402
from mvedit_amd.recon_loss import mesh_optim_loss  # noqa: E402
from mvedit_amd.tonemapping import Tonemapping, make_shading_fun  # noqa: E402
from scene import icosphere  # noqa: E402

gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_py.npz'))
S, nv = 64, 6
poses = torch.from_numpy(gold['poses'][:nv, :3].astype(np.float32)).cuda()
fl = S / (2 * np.tan(np.deg2rad(15)))
intr = torch.tensor([[fl, fl, S / 2, S / 2]], dtype=torch.float32).repeat(nv, 1).cuda()
v0, f = icosphere(3, 0.6)
f = torch.from_numpy(f).cuda()
mr = MeshRenderer(near=0.01, far=100)
lights = torch.nn.functional.normalize(torch.tensor([0.3, 0.5, 1.0], device='cuda'), dim=0).expand(nv, S, S, 3).contiguous()
shade = make_shading_fun(lights, 0.2, Tonemapping(device='cuda'))


def render(verts, shading=True):
    m = Mesh(verts, f, vc=torch.cat([torch.full_like(verts, 0.7), torch.ones_like(verts[:, :1])], -1))
    m.auto_normal()
    out = mr([m], poses[None], intr[None], S, S, shading_fun=shade if shading else None, normal_bg=[0.5, 0.5, 1.0])
    return m, out['rgba'][0], out['normal'][0], out['depth'][0]


with torch.no_grad():
    _, rgba_t, normal_t, _ = render(torch.from_numpy(v0).cuda())
tgt_m = rgba_t[..., 3:].contiguous()
tgt_rgb = (rgba_t[..., :3] / tgt_m.clamp(min=1e-3)).contiguous()
erode = -torch.nn.functional.max_pool2d(-tgt_m.permute(0, 3, 1, 2), 5, stride=1, padding=2).permute(0, 2, 3, 1).contiguous()
ys, xs = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing='ij')
dirs = torch.stack([(xs + 0.5 - S / 2) / fl, (ys + 0.5 - S / 2) / fl, torch.ones_like(xs)], -1)[None].repeat(nv, 1, 1, 1).cuda()
print('target: alpha mean', float(tgt_m.mean()), 'covered px/view', float((tgt_m > 0.5).float().sum() / nv))


def terms(verts):
    m, rgba, normal, depth = render(verts)
    res = mesh_optim_loss(rgba, normal, depth.detach(), tgt_rgb, erode, tgt_m, dirs, torch.ones(nv, device='cuda'), target_n=normal_t,
                          normal_reg_weight=1.0)
    lap, nc = mesh_regularizers(verts, f, m.face_normals)
    return dict(loss=res['loss'], alpha=res['alphas_loss'], rgb=res['pixel_rgb_loss'], lap=lap, nc=nc, alpha_sum=rgba[..., 3].sum())


base = torch.from_numpy(v0).cuda() * 0.75
g = torch.Generator().manual_seed(0)
rnd = torch.randn(3, 3, generator=g).cuda()
directions = {
    'scale': base / base.norm(dim=1, keepdim=True),
    'shift_x': torch.tensor([1.0, 0.0, 0.0], device='cuda').expand_as(base).contiguous(),
    'lowfreq': torch.sin(base @ rnd * 4.0),
}
for key in ('loss', 'alpha', 'rgb', 'lap', 'nc', 'alpha_sum'):
    v = base.clone().requires_grad_(True)
    t = terms(v)
    if not t[key].requires_grad:
        print(f'{key}: no grad'); continue
    gr, = torch.autograd.grad(t[key], v)
    for name, d in directions.items():
        ana = float((gr * d).sum())
        fds = []
        for eps in (2e-3, 5e-4):
            with torch.no_grad():
                lp, lm = float(terms(base + eps * d)[key]), float(terms(base - eps * d)[key])
            fds.append((lp - lm) / (2 * eps))
        print(f'{key:9s} {name:8s}: analytic {ana:+.5e}   finite-diff (2e-3, 5e-4) {fds[0]:+.5e} {fds[1]:+.5e}')

verts = base.clone().requires_grad_(True)
opt = torch.optim.Adam([verts], lr=4e-3)
for it in range(161):
    opt.zero_grad()
    t = terms(verts)
    (t['loss'] + 5.0 * (t['lap'] + t['nc'])).backward()
    opt.step()
    if it % 20 == 0:
        print(it, 'alpha %.4f rgb %.4f lap %.4f nc %.5f  mean radius %.4f' % (float(t['alpha']), float(t['rgb']), float(t['lap']), float(t['nc']),
                                                                             float(verts.detach().norm(dim=1).mean())))
