"""Writes tests/golden/triplane_ref.npz by EXECUTING the reference's `TriPlaneDecoder.point_decode` / `xyz_transform`
(lib/models/decoders/triplane_decoder.py:107-199) and `TriPlaneiNGPDecoder.point_decode` (lib/models/decoders/triplane_ingp_decoder.py:
142-212), cut out of the files with `ast`, over a stand-in `self` whose sub-modules are built the way the constructors build them (:58-96,
:62-116: nn.Linear stacks, SiLU, TruncExp from lib/ops/activation.py, Sigmoid).  Absent third-party pieces: the SH direction encoder is
lib/ops/shencoder (a native op; stood in for by oracle/sh_oracle.py, which tests/test_shencoder.py pins against scipy and against the
reference's own kernel) and the tiny-cuda-nn HashGrid (oracle/nerf_oracle.hashgrid_encode, UNPINNED as for iNGPDecoder).
Run from the repo root (needs /root/reference):  python tests/golden/make_triplane_golden.py"""
import ast
import importlib.util
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import nerf_oracle as NO  # noqa: E402
from oracle import sh_oracle as SH  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(HERE, 'triplane_ref.npz')
LOG2_HASHMAP = 12


def _methods(path, cls, names, ns):
    tree = ast.parse(open(path).read())
    c = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
    out = {}
    for fn in c.body:
        if isinstance(fn, ast.FunctionDef) and fn.name in names:
            exec(compile(ast.Module([fn], []), path, 'exec'), ns)
            out[fn.name] = ns[fn.name]
    return out


def build_self(C, hidden, hidden2, ingp, flip_z, plane_cfg, act, seed, n_levels=12, max_resolution=320):
    spec = importlib.util.spec_from_file_location('ref_activation', os.path.join(REF, 'lib/ops/activation.py'))
    A = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(A)
    torch.manual_seed(seed)
    s = types.SimpleNamespace()
    s.plane_cfg, s.interp_mode, s.flip_z, s.use_dir_enc, s.sigmoid_saturation = plane_cfg, 'bilinear', flip_z, True, 0.001
    s.code_dropout, s.scene_base, s.dir_net, s.bound = None, None, None, 1.0
    actl = dict(relu=nn.ReLU, silu=nn.SiLU, softplus=nn.Softplus)[act]
    s.base_net = nn.Sequential(nn.Linear(3 * C, hidden))
    s.base_activation = actl()
    s.density_net = nn.Sequential(nn.Linear(hidden, 1), A.TruncExp())
    s.color_net = nn.Sequential(nn.Linear(hidden + 16, hidden2), actl(), nn.Linear(hidden2, 3), nn.Sigmoid())
    s.dir_encoder = lambda d: torch.from_numpy(SH.sh_encode(d.detach().numpy(), 4)).float()
    for m in (s.base_net, s.density_net, s.color_net):
        for lin in m:
            if isinstance(lin, nn.Linear):
                nn.init.xavier_uniform_(lin.weight)
                nn.init.uniform_(lin.bias, -0.2, 0.2)
    if ingp:
        # a 2^12-row hash map keeps the fixture small (the reference fixes 2^19; the level table is data for the kernel either way)
        meta, rows = NO.grid_meta(n_levels, 16, max_resolution, 1.0, LOG2_HASHMAP)
        table = (torch.rand(rows, 2) * 2 - 1) * 0.5
        s.table = table
        s.encoder = lambda x01: torch.from_numpy(NO.hashgrid_encode(x01.detach().numpy(), table.numpy(), n_levels, max_resolution, 1.0, LOG2_HASHMAP))
        s.ingp_base_net = nn.Sequential(nn.Linear(2 * n_levels, hidden))
        nn.init.xavier_uniform_(s.ingp_base_net[0].weight)
        nn.init.uniform_(s.ingp_base_net[0].bias, -0.2, 0.2)
    return s


def main():
    out = {}
    cases = [('plain', dict(C=32, hidden=128, hidden2=128, ingp=False, flip_z=False, plane_cfg=['xy', 'xz', 'yz'], act='silu', seed=1)),
             ('flip', dict(C=8, hidden=64, hidden2=64, ingp=False, flip_z=True, plane_cfg=['xy', 'yz', 'xz'], act='relu', seed=2)),
             ('ingp', dict(C=32, hidden=128, hidden2=128, ingp=True, flip_z=False, plane_cfg=['xy', 'xz', 'yz'], act='silu', seed=3))]
    for tag, kw in cases:
        ns = dict(torch=torch, nn=nn, F=F)
        m = _methods(os.path.join(REF, 'lib/models/decoders/triplane_decoder.py'), 'TriPlaneDecoder', ('xyz_transform', 'point_decode'), ns)
        if kw['ingp']:
            m.update(_methods(os.path.join(REF, 'lib/models/decoders/triplane_ingp_decoder.py'), 'TriPlaneiNGPDecoder', ('point_decode',), ns))
        s = build_self(**kw)
        s.xyz_transform = types.MethodType(m['xyz_transform'], s)
        g = torch.Generator().manual_seed(11)
        N, hw = 600, 24
        code = torch.randn(1, 3, kw['C'], hw, hw + 8, generator=g)
        xyz = (torch.rand(N, 3, generator=g) * 2.3 - 1.15)                   # some points beyond the planes: border padding
        dirs = F.normalize(torch.randn(N, 3, generator=g), dim=-1)
        with torch.no_grad():
            sig, rgb, npts = m['point_decode'](s, [xyz], [dirs], code)
            sig_d, none_rgb, _ = m['point_decode'](s, [xyz], None, code, density_only=True)
        assert none_rgb is None and torch.equal(sig, sig_d) and npts == [N]
        out.update({f'{tag}_code': code.numpy(), f'{tag}_xyz': xyz.numpy(), f'{tag}_dirs': dirs.numpy(), f'{tag}_sigmas': sig.numpy(), f'{tag}_rgbs': rgb.numpy(),
                    f'{tag}_base_w': s.base_net[0].weight.detach().numpy(), f'{tag}_base_b': s.base_net[0].bias.detach().numpy(),
                    f'{tag}_dens_w': s.density_net[0].weight.detach().numpy(), f'{tag}_dens_b': s.density_net[0].bias.detach().numpy(),
                    f'{tag}_col1_w': s.color_net[0].weight.detach().numpy(), f'{tag}_col1_b': s.color_net[0].bias.detach().numpy(),
                    f'{tag}_col2_w': s.color_net[2].weight.detach().numpy(), f'{tag}_col2_b': s.color_net[2].bias.detach().numpy()})
        if kw['ingp']:
            out.update({f'{tag}_table': s.table.numpy(), f'{tag}_ingp_w': s.ingp_base_net[0].weight.detach().numpy(),
                        f'{tag}_ingp_b': s.ingp_base_net[0].bias.detach().numpy()})
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: v.shape for k, v in out.items() if k.endswith('rgbs')})


if __name__ == '__main__':
    main()
