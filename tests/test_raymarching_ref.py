"""Pins the oracle (and the HIP kernels) against the REFERENCE's own kernels.

oracle/_ref/_raymarching_ref*.so is the reference's lib/ops/raymarching extension built for gfx950
from /root/reference by oracle/build_ref.py (hipify-on-the-fly, -ffp-contract=off).  On the GPU box
it runs on the same MI355X; the reference sources are not needed at run time.

Everything the reference makes deterministic is compared bit-for-bit; ray offsets (atomicAdd order
in the reference, raymarching.cu:471) are compared as a permutation-invariant: per-ray sample
counts and per-ray sample contents.
"""
import os

import numpy as np
import pytest
import torch

from oracle import raymarching as O
from oracle import build_ref
from test_raymarching import small_case, _t, AABB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ref():
    mod = build_ref.load_module()
    if mod is None:
        pytest.skip('oracle/_ref not built (needs /root/reference at build time)')
    return mod


def test_ref_utils(ref, lib):
    from mvedit_amd import raymarching as G
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, (50000, 3)).astype(np.int32)
    idx = torch.empty(c.shape[0], dtype=torch.int32, device='cuda')
    ref.morton3D(_t(c), c.shape[0], idx)
    assert (idx.cpu().numpy() == O.morton3D(c)).all()
    assert torch.equal(idx, G.morton3D(_t(c)))
    back = torch.empty(c.shape[0], 3, dtype=torch.int32, device='cuda')
    ref.morton3D_invert(idx, c.shape[0], back)
    assert (back.cpu().numpy() == c).all()
    g = rng.random(64 ** 3).astype(np.float32)
    bits = torch.empty(g.size // 8, dtype=torch.uint8, device='cuda')
    ref.packbits(_t(g), g.size // 8, 0.5, bits)
    assert (bits.cpu().numpy() == O.packbits(g, 0.5)).all()
    cs = small_case(H=32, S=64, views=3)
    n = torch.empty(cs['o'].shape[0], device='cuda')
    f = torch.empty_like(n)
    ref.near_far_from_aabb(_t(cs['o']), _t(cs['d']), _t(AABB), cs['o'].shape[0], 0.2, n, f)
    assert (n.cpu().numpy() == cs['nears']).all() and (f.cpu().numpy() == cs['fars']).all()


@pytest.mark.parametrize('contract,dt_gamma,C,bound', [(False, 0.0, 1, 1.0), (True, 1.0 / 128, 2, 2.0)])
def test_ref_march_and_composite(ref, lib, contract, dt_gamma, C, bound):
    from mvedit_amd import raymarching as G
    cs = small_case(H=64, S=48, views=3, contract=contract, dt_gamma=dt_gamma, bound=bound, C=C)
    N = cs['o'].shape[0]
    o, d, bits = _t(cs['o']), _t(cs['d']), _t(cs['bits'])
    nears, fars, noises = _t(cs['nears']), _t(cs['fars']), _t(cs['noises'])
    rays = torch.empty(N, 2, dtype=torch.int32, device='cuda')
    counter = torch.zeros(1, dtype=torch.int32, device='cuda')
    ref.march_rays_train(o, d, bits, bound, contract, dt_gamma, 512, N, C, 64, nears, fars, None, None, None, rays,
                         counter, noises)
    M = int(counter.item())
    xyzs = torch.zeros(M, 3, device='cuda')
    dirs = torch.zeros(M, 3, device='cuda')
    ts = torch.zeros(M, 2, device='cuda')
    ref.march_rays_train(o, d, bits, bound, contract, dt_gamma, 512, N, C, 64, nears, fars, xyzs, dirs, ts, rays,
                         counter, noises)
    xo, do_, to, ro = O.march_rays_train(cs['o'], cs['d'], bound, cs['bits'], C, 64, cs['nears'], cs['fars'],
                                         cs['noises'], dt_gamma, 512, contract)
    rr = rays.cpu().numpy()
    assert M == xo.shape[0]
    assert (rr[:, 1] == ro[:, 1]).all(), 'per-ray sample counts differ from the reference kernels'
    xr, tr = xyzs.cpu().numpy(), ts.cpu().numpy()
    for r in np.nonzero(ro[:, 1] > 0)[0]:
        a, b = slice(rr[r, 0], rr[r, 0] + rr[r, 1]), slice(ro[r, 0], ro[r, 0] + ro[r, 1])
        assert (xr[a] == xo[b]).all() and (tr[a] == to[b]).all(), f'ray {r}: samples differ from the reference kernels'
    # compositing on the reference's own (atomic-ordered) ray table: HIP kernels vs reference kernels
    rng = np.random.default_rng(2)
    sig, rgb = _t((rng.random(M) * 8).astype(np.float32)), _t(rng.random((M, 3)).astype(np.float32))
    w_r = torch.zeros(M, device='cuda')
    ws_r, d_r, i_r = torch.empty(N, device='cuda'), torch.empty(N, device='cuda'), torch.empty(N, 3, device='cuda')
    ref.composite_rays_train_forward(sig, rgb, ts, rays, M, N, 1e-4, False, w_r, ws_r, d_r, i_r)
    w_h, ws_h, d_h, i_h = G.composite_rays_train(sig, rgb, ts, rays, 1e-4, False)
    for a, b in ((w_h, w_r), (ws_h, ws_r), (d_h, d_r), (i_h, i_r)):
        assert torch.equal(a, b), 'HIP compositing differs bitwise from the reference kernel on the same GPU'
    w_o, ws_o, d_o, i_o = O.composite_rays_train(sig.cpu().numpy(), rgb.cpu().numpy(), tr, rr, 1e-4, False)
    np.testing.assert_allclose(ws_r.cpu().numpy(), ws_o, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(i_r.cpu().numpy(), i_o, rtol=1e-5, atol=1e-6)
    # dump reference outputs as a golden fixture candidate (copied into tests/golden/ by hand)
    out = os.environ.get('MVE_DUMP_REF_GOLDEN')
    if out and not contract:
        np.savez_compressed(out, rays_counts=rr[:, 1], xyzs_sorted=xo, ts_sorted=to, ws=ws_r.cpu().numpy(),
                            img=i_r.cpu().numpy(), depth=d_r.cpu().numpy())
