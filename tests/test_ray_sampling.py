"""`ray_sample` / `get_raybatch_inds` mirrors (mvedit_amd/pipelines/ray_sampling.py) against the REFERENCE'S OWN METHODS executed here on the
CPU (lib/models/autoencoders/base_nerf.py:245-322, cut out of the class with `ast` because the module imports mmcv / mmgen): same seed ->
the same rays, patches and targets, bit for bit.  Where /root/reference is absent (GPU box) the committed golden holds the reference's
outputs (tests/golden/ray_sampling_ref.npz, written by this file's `python tests/test_ray_sampling.py`)."""
import ast
import os
import types

import numpy as np
import pytest
import torch

from mvedit_amd.pipelines.ray_sampling import get_raybatch_inds, ray_sample

REF = '/root/reference/lib/models/autoencoders/base_nerf.py'
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'ray_sampling_ref.npz')
CASES = [dict(S=1, I=4, h=16, w=24, ps=8, n=512), dict(S=2, I=3, h=16, w=16, ps=None, n=300), dict(S=1, I=2, h=8, w=8, ps=4, n=4096),
         dict(S=1, I=5, h=32, w=32, ps=16, n=1024)]


def _inputs(c, seed):
    g = torch.Generator().manual_seed(seed)
    shape = (c['S'], c['I'], c['h'], c['w'])
    return [torch.rand(*shape, 3, generator=g) for _ in range(3)], [torch.rand(*shape, 1, generator=g), torch.rand(*shape, 3, generator=g)]


def _run(fn_inds, fn_sample, c, seed):
    (ro, rd, img), extras = _inputs(c, seed)
    torch.manual_seed(seed + 100)
    inds, nb = fn_inds(img, c['n'])
    out = list(fn_sample(ro, rd, img, c['n'], None, extras))                        # draws its own permutation
    if inds is not None:
        out += list(fn_sample(ro, rd, img, c['n'], inds[1 % nb], extras))               # and with a batch of get_raybatch_inds
        out += [torch.stack(list(inds[:2]))] if len(inds) > 1 and inds[0].shape == inds[1].shape else [inds[0]]
    return [o.numpy() for o in out], nb


def _reference_methods(ps):
    cls = next(n for n in ast.parse(open(REF).read()).body if isinstance(n, ast.ClassDef) and n.name == 'BaseNeRF')
    ns = dict(torch=torch)
    for fn in cls.body:
        if isinstance(fn, ast.FunctionDef) and fn.name in ('ray_sample', 'get_raybatch_inds'):
            exec(compile(ast.Module([fn], []), REF, 'exec'), ns)
    me = types.SimpleNamespace(patch_loss=None if ps is None else object(), patch_size=ps)
    return (lambda imgs, n: ns['get_raybatch_inds'](me, imgs, n)), (lambda *a: ns['ray_sample'](me, *a))


def _mirror(ps):
    return (lambda imgs, n: get_raybatch_inds(imgs, n, ps)), (lambda ro, rd, img, n, inds, ex: ray_sample(ro, rd, img, n, inds, ex, ps))


@pytest.mark.parametrize('k', range(len(CASES)))
def test_mirror_equals_reference_methods(k):
    c = CASES[k]
    got, nb = _run(*_mirror(c['ps']), c, seed=k)
    if os.path.exists(REF):
        ref, nb_ref = _run(*_reference_methods(c['ps']), c, seed=k)
    else:
        g = np.load(GOLD)
        ref, nb_ref = [g[f'c{k}_{i}'] for i in range(int(g[f'c{k}_n']))], (int(g[f'c{k}_nb']) or None)
    assert nb == nb_ref and len(got) == len(ref)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and np.array_equal(a, b)


def test_patch_layout_is_what_the_loss_kernels_expect():
    """patches are (image, patch row, patch column)-major, pixels row-major inside: pixel (y, x) of patch (i, r, c) is image pixel
    (r * ps + y, c * ps + x) -- the layout recon_loss.nerf_optim_loss indexes as p = (n * ps + y) * ps + x"""
    I, h, w, ps = 2, 8, 12, 4
    img = torch.arange(I * h * w, dtype=torch.float32).reshape(1, I, h, w, 1).expand(-1, -1, -1, -1, 3).contiguous()
    _, _, tgt = ray_sample(img, img, img, I * h * w, patch_size=ps)
    n = 0
    for i in range(I):
        for r in range(h // ps):
            for c in range(w // ps):
                assert torch.equal(tgt[n, :, :, 0], img[0, i, r * ps:(r + 1) * ps, c * ps:(c + 1) * ps, 0])
                n += 1


if __name__ == '__main__':
    out = {}
    for k, c in enumerate(CASES):
        ref, nb = _run(*_reference_methods(c['ps']), c, seed=k)
        out.update({f'c{k}_{i}': a for i, a in enumerate(ref)})
        out[f'c{k}_n'], out[f'c{k}_nb'] = np.asarray(len(ref)), np.asarray(nb or 0)
    np.savez_compressed(GOLD, **out)
    print('wrote', GOLD, os.path.getsize(GOLD))
