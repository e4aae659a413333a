"""Gaussian blur / highpass (csrc/blur.hip + blur_core.h behind mvedit_amd.pipelines.utils.gaussian_blur / highpass) vs the torch
restatement of torchvision's gaussian_blur (oracle/blur_oracle.py; PARITY UNPINNED: torchvision is absent, the restatement follows its
published source) and vs an independent dense-matrix construction of the same operator.  Bar: 2e-6 absolute on [0, 1] images (fp32
summation order; the reference correlates with the 2-D outer-product kernel, the kernel here runs the two 1-D passes)."""
import numpy as np
import pytest
import torch

from oracle import blur_oracle as B


def _dense_1d(n, ksize, sigma):
    """[n, n] matrix of the reflect-padded 1-D correlation, built entry by entry (independent of both implementations)"""
    k = B.kernel1d(ksize, sigma, torch.float64)
    r = ksize // 2
    A = torch.zeros(n, n, dtype=torch.float64)
    for i in range(n):
        for t in range(ksize):
            j = i + t - r
            j = -j if j < 0 else (2 * (n - 1) - j if j >= n else j)
            A[i, j] += k[t]
    return A


@pytest.mark.parametrize('ksize,sigma', [(31, 5.0), (5, 1.2), (9, 3.0), (1, 1.0)])
def test_oracle_and_host_build_vs_dense_operator(ksize, sigma):
    from oracle import devcore as D
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 40, 36, generator=g, dtype=torch.float64)
    Ah, Aw = _dense_1d(40, ksize, sigma), _dense_1d(36, ksize, sigma)
    dense = torch.einsum('ij,ncjk,lk->ncil', Ah, x, Aw)
    assert (B.gaussian_blur(x, ksize, sigma) - dense).abs().max() < 1e-7            # the float32 kernel of torchvision vs float64 taps
    assert np.abs(D.gaussian_blur(x.numpy(), ksize, sigma) - dense.numpy()).max() < 2e-6
    gy = torch.rand(2, 3, 40, 36, generator=g, dtype=torch.float64)
    adj = torch.einsum('ij,ncik,kl->ncjl', Ah, gy, Aw)
    assert np.abs(D.gaussian_blur(gy.numpy(), ksize, sigma, adjoint=True) - adj.numpy()).max() < 2e-6
    # <A x, y> == <x, A^T y> for the host build itself
    lhs = (D.gaussian_blur(x.numpy(), ksize, sigma).astype(np.float64) * gy.numpy()).sum()
    rhs = (x.numpy() * D.gaussian_blur(gy.numpy(), ksize, sigma, adjoint=True).astype(np.float64)).sum()
    assert abs(lhs - rhs) < 1e-5 * abs(lhs)


def test_highpass_host_build_vs_oracle_and_its_autograd():
    from oracle import devcore as D
    g = torch.Generator().manual_seed(1)
    x = torch.rand(4, 3, 48, 48, generator=g, dtype=torch.float64).requires_grad_(True)
    hp = B.highpass(x)
    gy = torch.rand(hp.shape, generator=g, dtype=torch.float64)
    gx, = torch.autograd.grad((hp * gy).sum(), x)
    xn = x.detach().numpy()
    assert np.abs(D.gaussian_blur(xn, 31, 5.0, base=xn, offset=0.5) - hp.detach().numpy()).max() < 2e-6
    assert np.abs(D.gaussian_blur(gy.numpy(), 31, 5.0, adjoint=True, base=gy.numpy(), offset=0.0) - gx.numpy()).max() < 2e-6


@pytest.mark.gpu
def test_hip_blur_and_highpass_vs_oracle(lib):
    from mvedit_amd.pipelines.utils import gaussian_blur, highpass
    g = torch.Generator().manual_seed(2)
    for shape, ksize, sigma in (((8, 3, 128, 128), 31, 5.0), ((6, 1, 100, 76), 9, 3.0), ((2, 2, 33, 40), 5, 1.2)):
        x64 = torch.rand(*shape, generator=g, dtype=torch.float64).requires_grad_(True)
        gy = torch.rand(*shape, generator=g, dtype=torch.float64)
        y64 = B.gaussian_blur(x64, ksize, sigma)
        gx64, = torch.autograd.grad((y64 * gy).sum(), x64)
        x = x64.detach().float().cuda().requires_grad_(True)
        y = gaussian_blur(x, ksize, sigma)
        gx, = torch.autograd.grad((y * gy.float().cuda()).sum(), x)
        assert (y.detach().cpu().double() - y64.detach()).abs().max() < 2e-6 and (gx.cpu().double() - gx64).abs().max() < 2e-6
    x64 = torch.rand(8, 3, 128, 128, generator=g, dtype=torch.float64).requires_grad_(True)
    gy = torch.rand(8, 3, 128, 128, generator=g, dtype=torch.float64)
    hp64 = B.highpass(x64)
    gx64, = torch.autograd.grad((hp64 * gy).sum(), x64)
    x = x64.detach().float().cuda().requires_grad_(True)
    hp = highpass(x)
    gx, = torch.autograd.grad((hp * gy.float().cuda()).sum(), x)
    assert (hp.detach().cpu().double() - hp64.detach()).abs().max() < 2e-6 and (gx.cpu().double() - gx64).abs().max() < 2e-6
    # channels-last views as the loops pass them (`.permute(0, 3, 1, 2)`) go through .contiguous()
    xp = torch.rand(4, 64, 64, 3, device='cuda')
    assert torch.equal(highpass(xp.permute(0, 3, 1, 2)), highpass(xp.permute(0, 3, 1, 2).contiguous()))
