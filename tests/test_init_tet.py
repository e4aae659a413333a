"""NeRF -> DMTet hand-over (`mvedit_amd.pipelines.utils.init_tet`) against the REFERENCE'S OWN FUNCTION executed on the CPU
(lib/pipelines/utils.py:156-184, cut out with `ast`; its file download / np.load / device='cuda' are stood in for, the density comes from an
analytic stand-in decoder).  Golden for boxes without /root/reference: tests/golden/init_tet_ref.npz (`python tests/test_init_tet.py`)."""
import ast
import os
import types

import numpy as np
import torch

from mvedit_amd.pipelines.utils import init_tet
from scene import tet_grid

REF = '/root/reference/lib/pipelines/utils.py'
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'init_tet_ref.npz')


def _density(x):
    """an off-centre ellipsoid blob, density 30 inside falling to 0: occupied box is not the unit cube, some grid points leave [-1, 1]"""
    c = torch.tensor([0.15, -0.1, 0.05])
    r = (((x - c) / torch.tensor([0.55, 0.4, 0.7])) ** 2).sum(-1).sqrt()
    return (30.0 * (1.2 - r)).clamp(min=0)


class _Decoder:
    device = torch.device('cpu')

    def point_decode(self, xyzs, density_only=False):
        return _density(xyzs), None


def _grid():
    pos, tets = tet_grid(12)
    return (np.asarray(pos, np.float32) - 0.5) * -1.0, np.asarray(tets, np.int64)          # the file's convention: verts = -vertices * 2


def _reference(vertices, indices):
    fn = next(n for n in ast.parse(open(REF).read()).body if isinstance(n, ast.FunctionDef) and n.name == 'init_tet')

    class TorchProxy:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def tensor(data, dtype=None, device=None):
            return torch.tensor(data, dtype=dtype)
    ns = dict(torch=TorchProxy(), os=types.SimpleNamespace(path=types.SimpleNamespace(abspath=lambda p: p, join=lambda *a: '/'.join(a), exists=lambda p: True,
                                                                                   dirname=lambda p: p)),
              np=types.SimpleNamespace(load=lambda p: dict(vertices=vertices, indices=indices)), hf_hub_download=None, __file__='x')
    exec(compile(ast.Module([fn], []), REF, 'exec'), ns)
    model = types.SimpleNamespace(decoder=types.SimpleNamespace(point_density_decode=lambda xyzs, code: [_density(xyzs[0])]))
    return [t.numpy() for t in ns['init_tet'](model, None, density_thresh=5.0, resolution=128)]


def test_init_tet_equals_reference_function():
    vertices, indices = _grid()
    got = [t.numpy() for t in init_tet(_Decoder(), vertices, indices, density_thresh=5.0)]
    ref = _reference(vertices, indices) if os.path.exists(REF) else [np.load(GOLD)[k] for k in ('verts', 'indices', 'sdf')]
    assert got[1].dtype == np.int64 and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2])
    assert (got[2] == -1).any() and (got[2] > 0).any() and (np.abs(got[2]) < 1).any()      # outside the cube / inside the blob / the band


if __name__ == '__main__':
    v, i, s = _reference(*_grid())
    np.savez_compressed(GOLD, verts=v, indices=i, sdf=s)
    print('wrote', GOLD, os.path.getsize(GOLD))
