"""Spherical-harmonics direction encoder (csrc/sh.hip + sh_core.h behind mvedit_amd.shencoder) -- the reference's lib/ops/shencoder.
not gpu: the independent float64 oracle (oracle/sh_oracle.py) against scipy's complex harmonics on the unit sphere; the host build of the
         kernel source (recurrences) against the oracle, values and Jacobian, for every degree 1..8, on and off the sphere.
gpu    : HIP vs the oracle, and vs the REFERENCE'S OWN KERNEL rebuilt for gfx950 (oracle/_ref/_shencoder_ref*.so, oracle/build_ref.py).
Bars: fp32 evaluation of degree-7 polynomials: 3e-6 relative to the largest basis value of the batch (values reach ~3, Jacobian ~30)."""
import numpy as np
import pytest
import torch

from oracle import sh_oracle as S


def _points(n=400, seed=0):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    unit = v / np.linalg.norm(v, axis=1, keepdims=True)
    special = np.array([[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0], [0, 0, 0], [0.5, 0.5, 0.5]], np.float64)
    return np.concatenate([unit, special, v[:50] * 0.7]).astype(np.float32)             # on the sphere, poles / axes / origin, off the sphere


def test_oracle_against_scipy_on_the_unit_sphere():
    from scipy.special import sph_harm_y
    p = _points()[:400].astype(np.float64)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    theta, phi = np.arccos(np.clip(p[:, 2], -1, 1)), np.arctan2(p[:, 1], p[:, 0])
    out = S.sh_encode(p, 8)
    for l in range(8):
        for m in range(-l, l + 1):
            Y = sph_harm_y(l, abs(m), theta, phi)
            r = Y.real if m == 0 else (np.sqrt(2) * Y.real if m > 0 else np.sqrt(2) * Y.imag)
            assert np.abs(out[:, l * l + l + m] - r).max() < 1e-11, (l, m)
    # the first four in the closed form everyone knows: Y00, -c y, c z, -c x
    c = np.sqrt(3 / (4 * np.pi))
    assert np.allclose(out[:, :4], np.stack([np.full(400, 0.5 / np.sqrt(np.pi)), -c * p[:, 1], c * p[:, 2], -c * p[:, 0]], -1), atol=1e-14)


@pytest.mark.parametrize('degree', range(1, 9))
def test_host_build_of_kernel_source_vs_oracle(degree):
    from oracle import devcore as D
    p = _points()
    out64, jac64 = S.sh_encode(p, degree, jacobian=True)
    out, jac = D.sh_encode(p, degree, jacobian=True)
    assert np.abs(out - out64).max() <= 3e-6 * max(np.abs(out64).max(), 1.0)
    assert np.abs(jac - jac64).max() <= 3e-6 * max(np.abs(jac64).max(), 1.0)
    assert np.array_equal(D.sh_encode(p, degree), out)                               # values do not depend on asking for the Jacobian
    # Jacobian = derivative of the polynomial extension (central differences of the oracle)
    h = 1e-6
    for d in range(3):
        e = np.zeros(3)
        e[d] = h
        fd = (S.sh_encode(p.astype(np.float64) + e, degree) - S.sh_encode(p.astype(np.float64) - e, degree)) / (2 * h)
        assert np.abs(jac64[:, d] - fd).max() < 1e-6 * max(np.abs(fd).max(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize('degree', (1, 4, 8))
def test_hip_vs_oracle_and_reference_kernel(lib, degree):
    from mvedit_amd.shencoder import SHEncoder, sh_encode
    from oracle import build_ref
    p = _points(4000, seed=3)
    out64, jac64 = S.sh_encode(p, degree, jacobian=True)
    x = torch.from_numpy(p).cuda().requires_grad_(True)
    out = sh_encode(x, degree, True)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).cuda()
    gi, = torch.autograd.grad((out * g).sum(), x)
    gi64 = np.einsum('bc,bdc->bd', g.cpu().numpy().astype(np.float64), jac64)
    assert np.abs(out.detach().cpu().numpy() - out64).max() <= 3e-6 * max(np.abs(out64).max(), 1.0)
    assert np.abs(gi.cpu().numpy() - gi64).max() <= 1e-5 * max(np.abs(gi64).max(), 1.0)
    assert torch.equal(SHEncoder(degree=degree)(x.detach()[None] * 2, size=2)[0], sh_encode(x.detach() * 2 / 2, degree, False))
    ref = build_ref.load_module(build_ref.SH_NAME)
    if ref is None:
        pytest.skip('oracle/_ref/_shencoder_ref*.so not built (needs /root/reference at build time)')
    o_ref = torch.empty_like(out)
    j_ref = torch.empty(p.shape[0], 3 * degree * degree, device='cuda')
    ref.sh_encode_forward(x.detach(), o_ref, p.shape[0], 3, degree, j_ref)
    gi_ref = torch.zeros_like(x)
    ref.sh_encode_backward(g.contiguous(), x.detach(), p.shape[0], 3, degree, j_ref, gi_ref)
    print('max |ours - reference kernel|: values', float((out.detach() - o_ref).abs().max()), 'grad', float((gi - gi_ref).abs().max()))
    assert float((out.detach() - o_ref).abs().max()) <= 4e-6 * max(float(o_ref.abs().max()), 1.0)
    assert float((gi - gi_ref).abs().max()) <= 1e-5 * max(float(gi_ref.abs().max()), 1.0)
