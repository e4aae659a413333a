"""Writes tests/golden/bake_ref.npz by EXECUTING the reference's `MeshRenderer.bake_multiview`
(lib/models/decoders/mesh_renderer/base_mesh_renderer.py:507-603; the method is taken from the file with `ast`) together with the
reference's own get_ray_directions / depth_to_normal (lib/core/utils/geometry_utils.py) and edge_dilation (lib/ops/edge_dilation.py).
nvdiffrast is absent: the `dr` module the method calls is a stand-in built from this repo's rasteriser specification
(oracle/raster_oracle.c) and a differentiable torch bilinear fetch with wrap addressing -- so `torch.autograd.grad` of dr.texture w.r.t.
the dummy maps, which is how the reference obtains texel visibility, really is a gradient here.  What the file pins is the data flow
that oracle/bake_oracle.py restates: projection, depth, cosine weights, min-pool, visibility, accumulation, normalisation.
Run from the repo root (needs /root/reference):  python tests/golden/make_bake_golden.py"""
import ast
import importlib.util
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..'))
from oracle import raster as RO  # noqa: E402
from scene import face_atlas, icosphere  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(HERE, 'bake_ref.npz')
MIP_MAP = 128          # atlas size of the mip-mapped case (64^2 views of a small object: minified fetches on both sides)


def _fn(path, name, ns, cls=None):
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    node = [n for n in body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    exec(compile(ast.Module([node], []), path, 'exec'), ns)
    return ns[name]


def texture_torch(tex, uv):
    """bilinear, wrap addressing, texel centres at (i + 0.5) / n; differentiable w.r.t. tex.  tex [B,h,w,C], uv [B,...,2]."""
    B, h, w, C = tex.shape
    x, y = uv[..., 0] * w - 0.5, uv[..., 1] * h - 0.5
    fx, fy = torch.floor(x), torch.floor(y)
    wx1, wy1 = x - fx, y - fy
    out = 0
    b = torch.arange(B).view(B, *[1] * (uv.dim() - 2)).expand(uv.shape[:-1])
    for j, wy in ((0, 1 - wy1), (1, wy1)):
        for k, wx in ((0, 1 - wx1), (1, wx1)):
            iy, ix = torch.remainder(fy.long() + j, h), torch.remainder(fx.long() + k, w)
            out = out + (wx * wy)[..., None] * tex[b, iy, ix]
    return out


def dr_module(mip=False):
    """mip=True: rasterize / interpolate also return the pixel differentials and dr.texture honours filter_mode / uv_da through
    oracle/texture_mip_oracle.py (nvdiffrast's algorithm restated in torch, differentiable w.r.t. the texture)."""
    from oracle import texture_mip_oracle as TM
    dr = types.SimpleNamespace()

    def rasterize(glctx, pos, tri, resolution, grad_db=False):
        rast = torch.from_numpy(np.asarray(RO.rasterize(pos.detach().numpy(), tri.numpy(), tuple(resolution))))
        return rast, (TM.rasterize_db(pos.detach().float(), tri, rast) if mip else torch.zeros_like(rast))

    def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
        shared = attr.dim() == 2            # [V, A]: shared by all views
        a3 = attr[None] if shared else attr
        out = torch.from_numpy(np.asarray(RO.interpolate(a3.detach().numpy(), rast.numpy(), tri.numpy())))
        if mip and rast_db is not None and diff_attrs is not None:
            return out, TM.interpolate_da(attr.detach().float(), rast, rast_db, tri)
        return out, torch.zeros(*out.shape[:-1], 2 * out.shape[-1])

    def texture(tex, uv, uv_da=None, filter_mode=None):
        if mip and filter_mode == 'linear-mipmap-linear':
            return TM.texture(tex, uv, uv_da, filter_mode)
        return texture_torch(tex, uv)
    dr.rasterize, dr.interpolate, dr.texture = rasterize, interpolate, texture
    return dr


def scene(S=64, n_views=4, map_size=64):
    v, f = icosphere(2, 0.6)
    v = (v * (1 + 0.15 * np.sin(5 * v[:, :1]))).astype(np.float32)
    vt, ft = face_atlas(f)
    g = np.load(os.path.join(HERE, 'reference_py.npz'))
    poses = g['poses'][:n_views, :3].astype(np.float32)
    fl = S / (2 * np.tan(np.deg2rad(15)))
    intr = np.tile(np.array([[fl, fl, S / 2, S / 2]], np.float32), (n_views, 1))
    rng = np.random.default_rng(4)
    yy, xx = np.meshgrid(np.linspace(0, 1, S, dtype=np.float32), np.linspace(0, 1, S, dtype=np.float32), indexing='ij')
    images = np.stack([np.stack([0.5 + 0.5 * np.sin(7 * xx + i), yy, 0.5 + 0.5 * np.cos(9 * yy * xx + i)], -1) for i in range(n_views)])
    images = (images + rng.normal(0, 0.02, images.shape)).astype(np.float32)
    return v, f, vt, ft, poses, intr, images, S, map_size


def main():
    v, f, vt, ft, poses, intr, images, S, map_size = scene()
    gns = dict(torch=torch, F=F, np=np)
    for name in ('get_ray_directions', 'depth_to_normal'):
        _fn(os.path.join(REF, 'lib/core/utils/geometry_utils.py'), name, gns)
    spec = importlib.util.spec_from_file_location('ref_edge_dilation', os.path.join(REF, 'lib/ops/edge_dilation.py'))
    ed = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ed)
    ns = dict(torch=torch, F=F, dr=dr_module(), get_ray_directions=gns['get_ray_directions'], depth_to_normal=gns['depth_to_normal'],
              edge_dilation=ed.edge_dilation)
    bake = _fn(os.path.join(REF, 'lib/models/decoders/mesh_renderer/base_mesh_renderer.py'), 'bake_multiview', ns, cls='MeshRenderer')
    t = torch.from_numpy

    class Renderer:
        glctx, near, far, texture_filter = None, 0.01, 100.0, 'linear'
    mesh = types.SimpleNamespace(v=t(v), f=t(f), vt=t(vt), ft=t(ft), albedo=None, vc=None, textureless=True)
    # alpha = the mesh's own silhouette (the images being baked were rendered from this mesh)
    r = Renderer()
    alphas = []
    for i in range(poses.shape[0]):
        r_mat = np.concatenate([poses[i, :3, :1], -poses[i, :3, 1:3]], -1)
        proj = np.zeros((4, 4), np.float32)
        proj[0, 0], proj[0, 2] = 2 * intr[i, 0] / S, -2 * intr[i, 2] / S + 1
        proj[1, 1], proj[1, 2] = -2 * intr[i, 1] / S, -2 * intr[i, 3] / S + 1
        proj[2, 2], proj[2, 3], proj[3, 2] = -(r.far + r.near) / (r.far - r.near), -(2 * r.far * r.near) / (r.far - r.near), -1
        v_cam = (v - poses[i, :3, 3]) @ r_mat
        v_clip = np.concatenate([v_cam, np.ones_like(v_cam[:, :1])], -1) @ proj.T
        alphas.append((np.asarray(RO.rasterize(v_clip[None].astype(np.float32), f, (S, S)))[0, ..., 3:4] > 0).astype(np.float32))
    alphas = np.stack(alphas)
    (mesh,) = bake(r, [mesh], t(images)[None], t(alphas)[None], t(poses)[None], t(intr)[None], map_size=map_size, cos_weight_pow=8.0,
                   render_bs=3)
    albedo_linear = mesh.albedo.numpy()
    # the reference's DEFAULT filter: the same method over the mip-mapped stand-in (atlas twice as fine as the views can resolve, so that
    # both the visibility footprints and the image fetches really leave level 0)
    ns['dr'] = dr_module(mip=True)
    bake = _fn(os.path.join(REF, 'lib/models/decoders/mesh_renderer/base_mesh_renderer.py'), 'bake_multiview', ns, cls='MeshRenderer')
    r.texture_filter = 'linear-mipmap-linear'
    mesh2 = types.SimpleNamespace(v=t(v), f=t(f), vt=t(vt), ft=t(ft), albedo=None, vc=None, textureless=True)
    (mesh2,) = bake(r, [mesh2], t(images)[None], t(alphas)[None], t(poses)[None], t(intr)[None], map_size=MIP_MAP, cos_weight_pow=8.0,
                    render_bs=3)
    np.savez_compressed(OUT, alphas=alphas, albedo=albedo_linear, albedo_mip=mesh2.albedo.numpy())
    print('wrote', OUT, os.path.getsize(OUT), mesh.albedo.shape, float(mesh.albedo[..., :3].mean()), mesh2.albedo.shape,
          float(mesh2.albedo[..., :3].mean()))


if __name__ == '__main__':
    main()
