"""Writes tests/golden/mesh_normals_ref.npz by EXECUTING the reference's own `Mesh.auto_normal` (lib/models/decoders/mesh_renderer/
mesh_utils.py:359-382, the method body cut out of the class with `ast`: the module imports xatlas / trimesh / nvdiffrast) on the CPU in
float64, with torch autograd for the gradients of a random linear functional of (vn, face_normals) w.r.t. the vertices.
Run from the repo root (needs /root/reference):  python tests/golden/make_mesh_normals_golden.py"""
import ast
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import types

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_mesh_reg_golden import grid_patch, octa_sphere  # noqa: E402

REF = '/root/reference/lib/models/decoders/mesh_renderer/mesh_utils.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mesh_normals_ref.npz')


def load_auto_normal():
    cls = next(n for n in ast.parse(open(REF).read()).body if isinstance(n, ast.ClassDef) and n.name == 'Mesh')
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'auto_normal')
    ns = dict(torch=torch, F=F)
    exec(compile(ast.Module([fn], []), REF, 'exec'), ns)
    return ns['auto_normal']


def main():
    auto_normal = load_auto_normal()
    g = torch.Generator().manual_seed(0)
    out, i = {}, 0
    for name, (v, f) in (('sphere', octa_sphere(2)), ('patch', grid_patch(7))):
        v = (torch.from_numpy(v) + 0.03 * torch.randn(v.shape, generator=g, dtype=torch.float64)).requires_grad_(True)
        f = torch.from_numpy(f)[torch.randperm(f.shape[0], generator=g)].int()
        m = types.SimpleNamespace(v=v, f=f)
        auto_normal(m)
        a, b = torch.randn(m.vn.shape, generator=g, dtype=torch.float64), torch.randn(m.face_normals.shape, generator=g, dtype=torch.float64)
        g_v, = torch.autograd.grad((m.vn * a).sum() + (m.face_normals * b).sum(), v)
        out.update({f'c{i}_verts': v.detach().numpy(), f'c{i}_faces': f.numpy(), f'c{i}_vn': m.vn.detach().numpy(),
                    f'c{i}_face_normals': m.face_normals.detach().numpy(), f'c{i}_fn': m.fn.numpy(), f'c{i}_g_vn': a.numpy(), f'c{i}_g_fn': b.numpy(),
                    f'c{i}_g_verts': g_v.numpy()})
        print(name, 'V', v.shape[0], 'F', f.shape[0])
        i += 1
    out['n_cases'] = np.asarray(i)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
