"""Writes tests/golden/decoder_ref.npz by EXECUTING the reference's `iNGPDecoder.point_decode` / `density_blob` and its `MLP`
(lib/models/decoders/ingp_decoder.py:20-40, :100-120, taken from the file with `ast`) and `TruncExp` with its custom backward
(lib/ops/activation.py, imported: it only needs torch).  The hash-grid encoder is tiny-cuda-nn (absent): `self.encoder` is the oracle's
own restatement of it, so what this pins is everything AROUND the encoder: the (x + bound) / (2 bound) normalisation, the MLP, the
density blob, the truncated-exp density with its clamped gradient, the saturated sigmoid -- and the level-scale formula of the constructor.
Run from the repo root (needs /root/reference):  python tests/golden/make_decoder_golden.py"""
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
from oracle import nerf_oracle as N  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(HERE, 'decoder_ref.npz')


def points(n=3000, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, generator=g) * 2 - 1
    x[:200] *= 0.2                                     # inside the density blob's clamp radius
    return x


def main():
    path = os.path.join(REF, 'lib/models/decoders/ingp_decoder.py')
    tree = ast.parse(open(path).read())
    ns = dict(torch=torch, nn=nn, F=F, np=np)
    mlp_cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'MLP'][0]
    exec(compile(ast.Module([mlp_cls], []), path, 'exec'), ns)
    dec_cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'iNGPDecoder'][0]
    for fn in dec_cls.body:
        if isinstance(fn, ast.FunctionDef) and fn.name in ('point_decode', 'density_blob'):
            exec(compile(ast.Module([fn], []), path, 'exec'), ns)
    spec = importlib.util.spec_from_file_location('ref_activation', os.path.join(REF, 'lib/ops/activation.py'))
    act = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(act)

    params = N.make_nerf_params(seed=7, table_scale=0.5)
    mlp = ns['MLP'](24, 4, 64, 2, bias=True)
    mlp.net[0].weight.data, mlp.net[0].bias.data = torch.from_numpy(params['w1']), torch.from_numpy(params['b1']) + 0.05
    mlp.net[1].weight.data, mlp.net[1].bias.data = torch.from_numpy(params['w2']), torch.tensor([0.3, 0.1, -0.2, 0.05])
    params['b1'], params['b2'] = mlp.net[0].bias.data.numpy().copy(), mlp.net[1].bias.data.numpy().copy()
    dec = types.SimpleNamespace(bound=1.0, mlp=mlp, sigmoid_saturation=0.001, blob_density=1.0, blob_radius=0.2, sigma_activation=act.TruncExp())
    dec.encoder = lambda x01: torch.from_numpy(N.hashgrid_encode(x01.detach().numpy(), params['table'], 12, 320, 1.0))
    dec.density_blob = lambda x: ns['density_blob'](dec, x)
    x = points()
    with torch.no_grad():
        sigmas, rgbs, num = ns['point_decode'](dec, [x], None, None)
    assert num == [x.shape[0]]
    # the activation's backward: d sigma / d pre-activation = clamp(exp, 1e-6, 1e6)
    pre = torch.linspace(-20, 20, 81, requires_grad=True)
    act.trunc_exp(pre).sum().backward()
    out = dict(sigmas=sigmas.numpy(), rgbs=rgbs.numpy(), b1=params['b1'], b2=params['b2'], trunc_exp_pre=pre.detach().numpy(),
               trunc_exp_grad=pre.grad.numpy(),
               per_level_scale=np.float64(np.exp2(np.log2(320 * 1.0 / 16) / (12 - 1))))          # the constructor's expression, :64
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), float(sigmas.mean()), float(rgbs.mean()))


if __name__ == '__main__':
    main()
