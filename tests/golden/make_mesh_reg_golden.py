"""Writes tests/golden/mesh_reg_ref.npz by EXECUTING the reference's own mesh regularisers -- `compute_edge_to_face_mapping`,
`normal_consistency`, `laplacian_uniform`, `laplacian_smooth_loss` (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:20-101), cut out
of the file where it lies with `ast` (its module imports nvdiffrast) -- on the CPU (normal_consistency in float64; laplacian_smooth_loss in float32, the dtype its sparse matrix is built in), with torch
autograd for the gradients.
The functions call `.cuda()` on index tensors; for this run `torch.Tensor.cuda` is the identity.  Meshes: a closed subdivided octahedron
(every edge has two faces) and an open grid patch (boundary edges keep the reference's default face 0 on the missing side), both with
perturbed vertices.  Run from the repo root (needs /root/reference):  python tests/golden/make_mesh_reg_golden.py"""
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import ast
import os

import numpy as np
import torch

REF = '/root/reference/lib/models/decoders/mesh_renderer/base_mesh_renderer.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mesh_reg_ref.npz')
NAMES = {'compute_edge_to_face_mapping', 'normal_consistency', 'laplacian_uniform', 'laplacian_smooth_loss'}


def load_ref():
    ns = dict(torch=torch)
    for node in ast.parse(open(REF).read()).body:
        if isinstance(node, ast.FunctionDef) and node.name in NAMES:
            exec(compile(ast.Module([node], []), REF, 'exec'), ns)
    return ns


def octa_sphere(levels):
    v = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    v = [np.asarray(p, np.float64) for p in v]
    for _ in range(levels):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.stack(v), np.asarray(f, np.int64)


def grid_patch(n):
    ys, xs = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
    v = np.stack([xs.ravel() / (n - 1), ys.ravel() / (n - 1), 0.1 * np.sin(3.0 * xs.ravel() / n)], -1).astype(np.float64)
    f = []
    for y in range(n - 1):
        for x in range(n - 1):
            a, b, c, d = y * n + x, y * n + x + 1, (y + 1) * n + x, (y + 1) * n + x + 1
            f += [(a, b, d), (a, d, c)]
    return v, np.asarray(f, np.int64)


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    ref = load_ref()
    g = torch.Generator().manual_seed(0)
    out, i = {}, 0
    for name, (v, f) in (('sphere', octa_sphere(2)), ('patch', grid_patch(7))):
        v = torch.from_numpy(v) + 0.03 * torch.randn(v.shape, generator=g, dtype=torch.float64)
        f = torch.from_numpy(f)
        perm = torch.randperm(f.shape[0], generator=g)            # face order must not matter beyond the face-0 default
        f = f[perm]
        verts = v.float().requires_grad_(True)      # laplacian_uniform builds its matrix in float32: the reference's own precision
        fn = torch.nn.functional.normalize(torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=-1), dim=-1)
        fn = torch.nn.functional.normalize(fn + 0.2 * torch.randn(fn.shape, generator=g, dtype=torch.float64), dim=-1).requires_grad_(True)
        lap = ref['laplacian_smooth_loss'](verts, f.int())
        nc = ref['normal_consistency'](fn, f)
        g_v, = torch.autograd.grad(lap, verts)
        g_fn, = torch.autograd.grad(nc, fn)
        tpe = ref['compute_edge_to_face_mapping'](f)
        out.update({f'c{i}_verts': v.numpy(), f'c{i}_faces': f.numpy().astype(np.int32), f'c{i}_face_normals': fn.detach().numpy(),
                    f'c{i}_lap': lap.detach().numpy(), f'c{i}_nc': nc.detach().numpy(), f'c{i}_g_verts': g_v.numpy(), f'c{i}_g_fn': g_fn.numpy(),
                    f'c{i}_tris_per_edge': tpe.numpy()})
        print(name, 'V', v.shape[0], 'F', f.shape[0], 'E', tpe.shape[0], 'lap', float(lap), 'nc', float(nc))
        i += 1
    out['n_cases'] = np.asarray(i)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
