"""Marching tetrahedra (DMTet.__call__, base_mesh_renderer.py:104-188).

not gpu : the numpy restatement against tests/golden/dmtet_ref.npz, which holds the output of the REFERENCE class executed on CPU
          torch (tests/golden/make_dmtet_golden.py) -- bit for bit, vertex order and face order included.
gpu     : the sort-free HIP extraction against the same golden vectors and, on a 64^3 grid (1.57 M tets), against the restatement.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dmtet_oracle as DO
from scene import blob_sdf, tet_grid

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'dmtet_ref.npz')


@pytest.mark.parametrize('n,seed', [(10, 0), (16, 1)])
def test_oracle_matches_reference_output(n, seed):
    g = np.load(GOLD)
    pos, tets = tet_grid(n)
    v, f = DO.dmtet(pos, blob_sdf(pos, seed), tets)
    assert v.shape == g[f'verts_{n}'].shape and (v == g[f'verts_{n}']).all()
    assert (f == g[f'faces_{n}']).all()


def test_oracle_mesh_is_closed_manifold():
    pos, tets = tet_grid(12)
    v, f = DO.dmtet(pos, blob_sdf(pos, 3, noise=0.0), tets)
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all() and v.shape[0] - e.shape[0] // 2 + f.shape[0] == 2          # Euler characteristic of a sphere
    assert abs(np.linalg.norm(v, axis=-1).mean() - 0.6) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize('n,seed', [(10, 0), (16, 1)])
def test_gpu_matches_reference_output(lib, n, seed):
    from mvedit_amd.mesh_ops import DMTet
    g = np.load(GOLD)
    pos, tets = tet_grid(n)
    v, f = DMTet('cuda')(torch.from_numpy(pos), torch.from_numpy(blob_sdf(pos, seed)), torch.from_numpy(tets))
    assert f.dtype == torch.int64
    assert (v.cpu().numpy() == g[f'verts_{n}']).all() and (f.cpu().numpy() == g[f'faces_{n}']).all()


@pytest.mark.gpu
def test_gpu_large_grid_and_edge_cases(lib):
    from mvedit_amd.mesh_ops import DMTet
    dm = DMTet('cuda')
    pos, tets = tet_grid(64)
    sdf = blob_sdf(pos, 5, noise=0.03)
    v_o, f_o = DO.dmtet(pos, sdf, tets)
    tp, tt = torch.from_numpy(pos).cuda(), torch.from_numpy(tets).cuda()
    v, f = dm(tp, torch.from_numpy(sdf), tt)
    assert v_o.shape[0] > 30000 and (v.cpu().numpy() == v_o).all() and (f.cpu().numpy() == f_o).all()
    v2, f2 = dm(tp, torch.from_numpy(sdf), tt)                       # second call re-uses the converted tet grid
    assert torch.equal(v, v2) and torch.equal(f, f2)
    for fill in (1.0, -1.0):                                           # nothing crosses: empty mesh
        ve, fe = dm(tp, torch.full((pos.shape[0],), fill), tt)
        assert ve.shape == (0, 3) and fe.shape == (0, 3)
    # a single tetrahedron, every sign pattern: the 16 table rows
    p4 = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    t4 = torch.tensor([[0, 1, 2, 3]])
    for code in range(16):
        s = torch.tensor([1.0 if (code >> k) & 1 else -1.0 for k in range(4)]) * torch.tensor([0.3, 0.5, 0.7, 0.9])
        vo, fo = DO.dmtet(p4.numpy(), s.numpy(), t4.numpy())
        vh, fh = dm(p4, s, t4)
        assert (vh.cpu().numpy() == vo).all() and (fh.cpu().numpy() == fo).all(), code


@pytest.mark.gpu
@pytest.mark.parametrize('n,seed', [(10, 0), (16, 1)])
def test_gpu_backward_matches_reference_autograd(lib, n, seed):
    """d verts -> d pos, d sdf against the gradients the REFERENCE class produces through torch autograd (golden file)."""
    from mvedit_amd.mesh_ops import DMTet
    g = np.load(GOLD)
    pos, tets = tet_grid(n)
    tp = torch.from_numpy(pos).cuda().requires_grad_(True)
    ts = torch.from_numpy(blob_sdf(pos, seed)).cuda().requires_grad_(True)
    v, f = DMTet('cuda')(tp, ts, torch.from_numpy(tets))
    assert v.requires_grad and (v.detach().cpu().numpy() == g[f'verts_{n}']).all() and (f.cpu().numpy() == g[f'faces_{n}']).all()
    R = torch.from_numpy(np.random.default_rng(7).standard_normal(tuple(v.shape)).astype(np.float32)).cuda()
    (v * R).sum().backward()
    for got, want, name in ((tp.grad, g[f'grad_pos_{n}'], 'pos'), (ts.grad, g[f'grad_sdf_{n}'], 'sdf')):
        want_t = torch.from_numpy(want)
        scale = want_t.abs().max().item()
        assert scale > 0 and (got.cpu() - want_t).abs().max().item() <= 1e-5 * scale, name      # float atomics: summation order only
    # no grad requested -> plain tensors
    v2, _ = DMTet('cuda')(tp.detach(), ts.detach(), torch.from_numpy(tets))
    assert not v2.requires_grad
