"""Geometry gradients of the mesh rasteriser (mve_rasterize_backward, mve_interpolate_backward_rast behind the autograd Functions of
mvedit_amd.mesh_ops) vs torch autograd over the continuous barycentric model (oracle/raster_grad_oracle.py).
SURVEY section 8(f) rank 1 ("raster / interpolate backward").  nvdiffrast is absent: the differentiated forward is this repo's
rasteriser specification; what is checked is that the hand-derived chain rule equals autograd and that a vertex fit converges."""
import math

import numpy as np
import pytest
import torch

from oracle import raster as RO
from oracle import raster_grad_oracle as R
from scene import icosphere


def _scene(H=24, W=32):
    v, f = icosphere(2, 0.6)
    pos = torch.from_numpy(v).float()

    def clip(ang):
        c, s = math.cos(ang), math.sin(ang)
        rot = torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=torch.float32)
        p = pos @ rot.T
        z = p[:, 2] + 2.5
        return torch.stack([p[:, 0] * 2.0, p[:, 1] * 2.0, (z - 2.5) * 0.5, z], dim=-1)

    P = torch.stack([clip(0.3), clip(1.1)])
    tri = torch.from_numpy(f.astype(np.int32))
    return P, tri, H, W


# ------------------------------------------------------------------------------------------------ CPU
def test_closed_form_equals_autograd():
    """The chain rule the kernel implements (restated in the oracle, operation for operation) against autograd, in float64."""
    P, tri, H, W = _scene()
    rast = torch.from_numpy(np.asarray(RO.rasterize(P.numpy(), tri.numpy(), (H, W))))
    ids = rast[..., 3].long() - 1
    b_idx, yy, xx = torch.nonzero(ids >= 0, as_tuple=True)
    assert len(b_idx) > 200
    Pd = P.double().requires_grad_(True)
    u, v, z = R.rast_continuous(Pd, tri, b_idx, ids[b_idx, yy, xx], xx, yy, H, W)
    # the continuous model is the rasteriser's output up to its 1/256-pixel vertex snapping
    assert (u.detach().float() - rast[b_idx, yy, xx, 0]).abs().max() < 2e-2 and (z.detach().float() - rast[b_idx, yy, xx, 2]).abs().max() < 2e-3
    g = torch.randn(2, H, W, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    (u * g[b_idx, yy, xx, 0] + v * g[b_idx, yy, xx, 1] + z * g[b_idx, yy, xx, 2]).sum().backward()
    gp = R.rasterize_backward(P.double(), tri, rast.double(), g)
    assert (gp - Pd.grad).abs().max() <= 1e-12 * Pd.grad.abs().max()
    # interpolate's (u, v) gradient
    attr = torch.randn(1, P.shape[1], 5, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    rr = rast.double().clone().requires_grad_(True)
    a = attr[0][tri[ids.clamp(min=0)].long()]
    out = (rr[..., 0:1] * a[..., 0, :] + rr[..., 1:2] * a[..., 1, :] + (1 - rr[..., 0:1] - rr[..., 1:2]) * a[..., 2, :]) * (ids >= 0)[..., None]
    go = torch.randn(out.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    (out * go).sum().backward()
    assert (R.interpolate_backward_rast(attr, tri, rast.double(), go)[..., :2] - rr.grad[..., :2]).abs().max() < 1e-12


def _aa_case():
    P, tri, H, W = _scene()
    Pn, trin = P.numpy(), tri.numpy()
    rast = np.asarray(RO.rasterize(Pn, trin, (H, W)))
    opp = RO.edge_opposites(trin)
    rng = np.random.default_rng(0)
    color = rng.random((2, H, W, 3), dtype=np.float32)
    G = rng.standard_normal((2, H, W, 3)).astype(np.float32)
    return Pn, trin, rast, opp, color, G


def test_antialias_position_gradient_vs_finite_differences():
    """The silhouette gradient: (1) the Python restatement of the pair rule reproduces the C oracle's antialias bit for bit, so it
    reports the same discrete choices; (2) the closed-form gradient built on those choices equals central finite differences of the
    C oracle's forward w.r.t. every vertex coordinate that carries gradient."""
    Pn, tri, rast, opp, color, G = _aa_case()
    assert np.array_equal(RO.antialias(color, rast, Pn, tri, opp), R.antialias_forward(color, rast, Pn, tri, opp))
    gp = R.antialias_backward_pos(color, rast, Pn, tri, opp, G)
    idx = np.argwhere(np.abs(gp) > 1e-6)
    assert len(idx) >= 20
    loss = lambda Pp: float((RO.antialias(color, rast, Pp.astype(np.float32), tri, opp).astype(np.float64) * G).sum())
    bad = 0
    for b, vi, c in idx:
        Pp, Pm = Pn.astype(np.float64).copy(), Pn.astype(np.float64).copy()
        Pp[b, vi, c] += 2e-3
        Pm[b, vi, c] -= 2e-3
        fd = (loss(Pp) - loss(Pm)) / 4e-3
        bad += abs(fd - gp[b, vi, c]) > 5e-2 * max(abs(gp[b, vi, c]), 1e-3)
    assert bad <= len(idx) // 20            # a perturbation may flip a discrete choice now and then


def test_device_source_on_host_equals_closed_forms():
    """The per-pixel bodies of the three HIP kernels (mvedit_amd/csrc/raster_grad_core.h) compiled for the HOST (oracle/devcore_host.cpp,
    fp32, sequential) against the float64 closed forms above: the device arithmetic itself is checked here; what is left for the
    first GPU run is the launch geometry and the atomics."""
    from oracle import devcore as D
    P, tri, H, W = _scene()
    Pn, trin = P.numpy(), tri.numpy()
    rast = np.asarray(RO.rasterize(Pn, trin, (H, W)))
    rng = np.random.default_rng(3)
    g_rast = rng.standard_normal(rast.shape).astype(np.float32)
    want = R.rasterize_backward(P.double(), tri, torch.from_numpy(rast).double(), torch.from_numpy(g_rast).double()).numpy()
    got = D.rasterize_backward(Pn, trin, rast, g_rast)
    assert np.abs(got - want).max() < 5e-5 * np.abs(want).max()
    attr = rng.standard_normal((1, Pn.shape[1], 5)).astype(np.float32)
    go = rng.standard_normal((2, H, W, 5)).astype(np.float32)
    want = R.interpolate_backward_rast(torch.from_numpy(attr).double(), tri, torch.from_numpy(rast).double(), torch.from_numpy(go).double()).numpy()
    assert np.abs(D.interpolate_backward_rast(attr, rast, trin, go) - want).max() < 1e-5
    Pn, trin, rast, opp, color, G = _aa_case()
    want = R.antialias_backward_pos(color, rast, Pn, trin, opp, G)
    got = D.antialias_backward_pos(color, rast, Pn, trin, opp, G)
    assert (np.abs(got) > 0).sum() >= 20 and np.abs(got - want).max() < 1e-5 * np.abs(want).max()


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_rasterize_and_interpolate_geometry_gradient_vs_oracle(lib):
    from mvedit_amd.mesh_ops import interpolate, rasterize
    P, tri, H, W = _scene()
    attr = torch.randn(1, P.shape[1], 5, generator=torch.Generator().manual_seed(1))
    go = torch.randn(2, H, W, 5, generator=torch.Generator().manual_seed(2))
    gz = torch.randn(2, H, W, generator=torch.Generator().manual_seed(3))
    pg = P.cuda().requires_grad_(True)
    rast = rasterize(pg, tri.cuda(), (H, W))
    out = interpolate(attr.cuda(), rast, tri.cuda())
    ((out * go.cuda()).sum() + (rast[..., 2] * gz.cuda()).sum()).backward()
    # oracle: same upstream gradients through the closed form (float64), on the rast the GPU produced
    r = rast.detach().cpu().double()
    g_rast = R.interpolate_backward_rast(attr.double(), tri, r, go.double())
    g_rast[..., 2] = gz.double() * (r[..., 3] > 0)
    want = R.rasterize_backward(P.double(), tri, r, g_rast)
    got = pg.grad.cpu().double()
    assert ((got - want).norm() / want.norm()) < 1e-4 and (got - want).abs().max() < 1e-3 * want.abs().max()


@pytest.mark.gpu
def test_vertex_fit_through_depth_converges(lib):
    """Recover a per-vertex radial scale from a target z/w image by gradient descent through rasterize (interior gradients only:
    the silhouette term of antialias is not part of this round)."""
    from mvedit_amd.mesh_ops import rasterize
    P, tri, H, W = _scene(48, 48)
    tgt = rasterize(P.cuda(), tri.cuda(), (H, W))[..., 2].detach()
    scale = torch.full((P.shape[1], 1), 1.04, device='cuda', requires_grad=True)
    opt = torch.optim.Adam([scale], lr=5e-3)
    losses = []
    for _ in range(60):
        pos = P.cuda().clone()
        pos = torch.cat([pos[..., :2] * scale, pos[..., 2:]], dim=-1)
        rast = rasterize(pos, tri.cuda(), (H, W))
        m = ((rast[..., 3] > 0) & (tgt != 0)).float()
        loss = (((rast[..., 2] - tgt) * m) ** 2).sum() / m.sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert torch.isfinite(scale).all() and losses[-1] < losses[0]


@pytest.mark.gpu
def test_antialias_position_gradient_vs_oracle(lib):
    from mvedit_amd.mesh_ops import antialias
    Pn, tri, rast, opp, color, G = _aa_case()
    pg = torch.from_numpy(Pn).cuda().requires_grad_(True)
    cg = torch.from_numpy(color).cuda().requires_grad_(True)
    out = antialias(cg, torch.from_numpy(rast).cuda(), pg, torch.from_numpy(tri).cuda(), torch.from_numpy(opp).cuda())
    assert np.array_equal(out.detach().cpu().numpy(), RO.antialias(color, rast, Pn, tri, opp))
    (out * torch.from_numpy(G).cuda()).sum().backward()
    want = R.antialias_backward_pos(color, rast, Pn, tri, opp, G)
    got = pg.grad.cpu().double().numpy()
    assert np.abs(got - want).max() < 1e-4 * np.abs(want).max() and cg.grad is not None
