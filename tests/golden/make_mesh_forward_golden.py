"""Writes tests/golden/mesh_forward_ref.npz by EXECUTING the reference's `MeshRenderer.forward`
(lib/models/decoders/mesh_renderer/base_mesh_renderer.py:207-395, single-scene branch; with its own interpolate_hwc and
lib/ops/edge_dilation.py) over the stand-in `dr` module of make_bake_golden.py extended by dr.antialias (the raster oracle's).  The
projected vertices the method hands to dr.rasterize are recorded, so that the restatement (oracle/mesh_forward_oracle.py) can be run on
bit-identical input.  Run from the repo root (needs /root/reference):  python tests/golden/make_mesh_forward_golden.py"""
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
OUT = os.path.join(HERE, 'mesh_forward_ref.npz')
spec = importlib.util.spec_from_file_location('make_bake_golden', os.path.join(HERE, 'make_bake_golden.py'))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def scene(S=64, nv=2):
    v, f = icosphere(3, 0.6)
    vn = (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)
    vt, ft = face_atlas(f)
    tex = np.random.default_rng(1).random((64, 64, 4)).astype(np.float32)
    vcol = np.concatenate([np.random.default_rng(2).random((v.shape[0], 3)), np.full((v.shape[0], 1), 0.9)], -1).astype(np.float32)
    g = np.load(os.path.join(HERE, 'reference_py.npz'))
    poses = g['poses'][:nv, :3].astype(np.float32)
    fl = S / (2 * np.tan(np.deg2rad(15)))
    intr = np.tile(np.array([[fl, fl, S / 2, S / 2]], np.float32), (nv, 1))
    return v, f, vn, vt, ft, tex, vcol, poses, intr, S


def shade(world_pos, albedo, world_normal, fg_mask):
    return albedo * 0.5 + 0.1 * (world_normal[..., :1] * 0.5 + 0.5) + 0.05 * world_pos[..., 1:2]


def main():
    v, f, vn, vt, ft, tex, vcol, poses, intr, S = scene()
    dr = B.dr_module()
    recorded = {}
    base_rasterize = dr.rasterize

    def rasterize(glctx, pos, tri, resolution, grad_db=False):
        recorded['v_clip'] = pos.detach().numpy().copy()
        return base_rasterize(glctx, pos, tri, resolution, grad_db)

    def antialias(color, rast, pos, tri):
        return torch.from_numpy(np.asarray(RO.antialias(color.detach().numpy(), rast.numpy(), pos.detach().numpy(), tri.numpy())))
    dr.rasterize, dr.antialias = rasterize, antialias
    spec2 = importlib.util.spec_from_file_location('ref_edge_dilation', os.path.join(REF, 'lib/ops/edge_dilation.py'))
    ed = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(ed)
    path = os.path.join(REF, 'lib/models/decoders/mesh_renderer/base_mesh_renderer.py')
    ns = dict(torch=torch, F=F, dr=dr, edge_dilation=ed.edge_dilation)
    B._fn(path, 'interpolate_hwc', ns)
    forward = B._fn(path, 'forward', ns, cls='MeshRenderer')
    t = torch.from_numpy
    out = {}
    tex_big = np.random.default_rng(5).random((256, 256, 4)).astype(np.float32)     # 256^2 atlas seen at 64^2: minified 2-4x
    out['tex_big'] = tex_big
    for tag, ssaa, kw in (('tex_aa', 1, {}), ('tex_aa_ssaa2', 2, {}), ('vc_shade_dilate', 1, dict(shading_fun=shade, dilate_edges=1, aa=False)),
                          ('texmip_aa', 1, {}), ('texmip_aa_ssaa2', 2, {})):
        mip = tag.startswith('texmip')
        if mip:             # the reference's default filter over the mip-mapped stand-in (make_bake_golden.dr_module(mip=True))
            dr2 = B.dr_module(mip=True)
            base2 = dr2.rasterize

            def rasterize2(glctx, pos, tri, resolution, grad_db=False, _b=base2):
                recorded['v_clip'] = pos.detach().numpy().copy()
                return _b(glctx, pos, tri, resolution, grad_db)
            dr2.rasterize, dr2.antialias = rasterize2, antialias
            ns2 = dict(torch=torch, F=F, dr=dr2, edge_dilation=ed.edge_dilation)
            B._fn(path, 'interpolate_hwc', ns2)
            forward_fn = B._fn(path, 'forward', ns2, cls='MeshRenderer')
        else:
            forward_fn = forward
        r = types.SimpleNamespace(glctx=None, near=0.01, far=100.0, texture_filter='linear-mipmap-linear' if mip else 'linear', ssaa=ssaa)
        if mip:
            mesh = types.SimpleNamespace(v=t(v), f=t(f), vn=t(vn), fn=t(f), vt=t(vt), ft=t(ft), albedo=t(tex_big), vc=None)
        elif tag.startswith('tex'):
            mesh = types.SimpleNamespace(v=t(v), f=t(f), vn=t(vn), fn=t(f), vt=t(vt), ft=t(ft), albedo=t(tex), vc=None)
        else:
            mesh = types.SimpleNamespace(v=t(v), f=t(f), vn=t(vn), fn=t(f), vt=None, ft=None, albedo=None, vc=t(vcol))
        with torch.no_grad():
            res = forward_fn(r, [mesh], t(poses)[None], t(intr)[None], S, S, **kw)
        out[f'{tag}_rgba'], out[f'{tag}_depth'], out[f'{tag}_normal'] = (res[k][0].detach().numpy() for k in ('rgba', 'depth', 'normal'))
        out[f'{tag}_v_clip'] = recorded['v_clip']
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: v.shape for k, v in out.items() if k.endswith('rgba')})


if __name__ == '__main__':
    main()
