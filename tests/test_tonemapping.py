"""Tone-mapping table and render-step shading (csrc/shading.hip behind mvedit_amd.tonemapping) vs outputs of the reference's own
Tonemapping class (tests/golden/tonemap_ref.npz, written by tests/golden/make_tonemap_golden.py).
Bars: table interpolation in 'log' mode is the reference's expression op by op (no fma): bit-exact; 'linear' modes pass through
log2 / exp2 whose device implementations differ from the host's in the last place: 2e-6 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import tonemap_oracle as T

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tonemap_ref.npz'))
t = lambda k: torch.from_numpy(G[k])


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_and_host_tables_match_reference_output():
    from mvedit_amd.tonemapping import Tonemapping
    lx, ly = T.tables()
    assert np.array_equal(lx.numpy(), G['lut_x']) and np.array_equal(ly.numpy(), G['lut_y'])
    tm = Tonemapping(device='cpu')
    assert np.array_equal(tm.lut_x.numpy(), G['lut_x']) and np.array_equal(tm.lut_y.numpy(), G['lut_y'])
    assert np.array_equal(T.lut(lx, ly, t('x_log')).numpy(), G['lut_log'])
    assert np.array_equal(T.inverse_lut(lx, ly, t('y')).numpy(), G['inv_log'])
    np.testing.assert_allclose(T.lut(lx, ly, t('x_lin'), 'linear').numpy(), G['lut_lin'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(T.inverse_lut(lx, ly, t('y'), 'linear').numpy(), G['inv_lin'], rtol=1e-6)
    np.testing.assert_allclose(T.shade_views(t('rgba'), t('normal_fg'), t('cam_lights'), 0.1, 1.0, lx, ly).numpy(), G['shaded_tm'], rtol=1e-6, atol=1e-7)
    # smooth_forward of the mirror is the curve the knots sit on
    assert torch.equal(tm.smooth_forward(tm.lut_x), tm.lut_y)


def test_lut_round_trip_property():
    """inverse_lut(lut(x)) == x inside the table's range (both maps are piecewise linear over the same knots)."""
    lx, ly = T.tables()
    x = torch.linspace(-8.9, 2.9, 1001)
    assert (T.inverse_lut(lx, ly, T.lut(lx, ly, x)) - x).abs().max() < 2e-5


def test_device_source_on_host_matches_reference_output():
    """mvedit_amd/csrc/shading_core.h -- the arithmetic of mve_tonemap_lut / mve_shade_views -- compiled for the HOST
    (oracle/devcore_host.cpp) against outputs of the reference's Tonemapping class: bit-exact in 'log' mode and on the shaded batch;
    last-place differences of log2f / exp2f in the 'linear' modes."""
    from oracle import devcore as D
    lx, ly = G['lut_x'], G['lut_y']
    assert np.array_equal(D.tonemap_lut(G['x_log'], lx, ly), G['lut_log'])
    assert np.array_equal(D.tonemap_lut(G['y'], lx, ly, inverse=True), G['inv_log'])
    np.testing.assert_allclose(D.tonemap_lut(G['x_lin'], lx, ly, linear=True), G['lut_lin'], rtol=1e-6, atol=2e-7)
    np.testing.assert_allclose(D.tonemap_lut(G['y'], lx, ly, inverse=True, linear=True), G['inv_lin'], rtol=1e-6)
    np.testing.assert_allclose(D.shade_views(G['rgba'], G['normal_fg'], G['cam_lights'], 0.1, 1.0, lx, ly), G['shaded_tm'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(D.shade_views(G['rgba'], G['normal_fg'], G['cam_lights'], 0.1, 1.0), G['shaded_plain'], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_lut_and_inverse_vs_reference_output(lib):
    from mvedit_amd.tonemapping import Tonemapping
    tm = Tonemapping()
    assert torch.equal(tm.lut(t('x_log').cuda()).cpu(), t('lut_log'))
    assert torch.equal(tm.inverse_lut(t('y').cuda()).cpu(), t('inv_log'))
    np.testing.assert_allclose(tm.lut(t('x_lin').cuda(), 'linear').cpu().numpy(), G['lut_lin'], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(tm.inverse_lut(t('y').cuda(), 'linear').cpu().numpy(), G['inv_lin'], rtol=2e-6)
    h = tm.lut(t('x_log').half().cuda())
    assert h.dtype == torch.float16 and h.shape == t('x_log').shape                 # dtype round trip as the reference's `.to(dtype)`


@pytest.mark.gpu
def test_shade_views_vs_reference_output(lib):
    from mvedit_amd.tonemapping import Tonemapping, shade_views
    rgba, nf, lights = t('rgba').cuda(), t('normal_fg').cuda(), t('cam_lights').cuda()
    out = shade_views(rgba, nf, lights, 0.1, 1.0, Tonemapping())
    assert out.shape == (1, 3, 20, 24, 3)
    # log2 of the shading term and the dot product's summation order are the only non-identical steps
    np.testing.assert_allclose(out.cpu().numpy(), G['shaded_tm'], rtol=1e-5, atol=2e-6)
    plain = shade_views(rgba, nf, lights, 0.1, 1.0, None)
    np.testing.assert_allclose(plain.cpu().numpy(), G['shaded_plain'], rtol=1e-6, atol=1e-6)
    # 6 x 512^2 pixels, the production batch: finite, and equal to the small-batch result on the overlapping view
    big = torch.rand(1, 6, 512, 512, 4, device='cuda')
    nfb = torch.rand(1, 6, 512, 512, 3, device='cuda')
    lb = torch.nn.functional.normalize(torch.randn(6, 3, device='cuda'), dim=-1)
    full = shade_views(big, nfb, lb, 0.1, 1.0, Tonemapping())
    assert torch.isfinite(full).all() and torch.equal(full[:, 2:3], shade_views(big[:, 2:3], nfb[:, 2:3], lb[2:3], 0.1, 1.0, Tonemapping()))


def test_lut_gradient_host_build_vs_reference_autograd():
    """lut / inverse_lut sit inside the optimisation loops (mvedit_3d_pipeline.py:419-420, :568-571): their gradient must be the one torch
    autograd gives the reference's expressions (golden: float64 autograd over the reference class).  Host build of sh_lut_grad, 2e-6."""
    from oracle import devcore as D
    for key, x, inv, lin in (('log', G['x_log'][:4000], 0, 0), ('lin', G['x_lin'], 0, 1), ('inv_log', G['y'][:4000], 1, 0), ('inv_lin', G['y'][:4000], 1, 1)):
        got = D.tonemap_lut_grad(x, G['lut_x'], G['lut_y'], inv, lin)
        ref = G['grad_' + key]
        assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max(), key


@pytest.mark.gpu
def test_hip_lut_is_differentiable_like_the_reference():
    from mvedit_amd.tonemapping import Tonemapping
    tm = Tonemapping(device='cuda')
    for key, x, fn in (('log', G['x_log'][:4000], lambda v: tm.lut(v)), ('lin', G['x_lin'], lambda v: tm.lut(v, input_mode='linear')),
                       ('inv_log', G['y'][:4000], lambda v: tm.inverse_lut(v)), ('inv_lin', G['y'][:4000], lambda v: tm.inverse_lut(v, output_mode='linear'))):
        leaf = torch.from_numpy(x).float().cuda().requires_grad_(True)
        gr, = torch.autograd.grad(fn(leaf).sum(), leaf)
        ref = G['grad_' + key]
        assert np.abs(gr.cpu().numpy() - ref).max() <= 3e-6 * np.abs(ref).max(), key
    # the shading expression of the loops, end to end through autograd: lut(inverse_lut(albedo) + log2(shading))
    albedo = torch.rand(1000, 3, device='cuda').requires_grad_(True)
    shading = (torch.rand(1000, 1, device='cuda') + 0.2).requires_grad_(True)
    out = tm.lut(tm.inverse_lut(albedo) + shading.clamp(min=1e-6).log2())
    ga, gs = torch.autograd.grad(out.sum(), (albedo, shading))
    assert torch.isfinite(ga).all() and torch.isfinite(gs).all() and float(ga.abs().max()) > 0 and float(gs.abs().max()) > 0


# ================================================================================================ shading functions of the mesh path
GS = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'shading_fun_ref.npz'))


def test_shade_points_host_build_vs_reference_shading_funs():
    """sh_shade_point (the body of mve_shade_points, forward and backward) against the reference's own `make_shading_fun` /
    `make_nerf_shading_fun` executed in float64 with autograd (tests/golden/make_shading_fun_golden.py): 3e-6 of each tensor's scale."""
    from oracle import devcore as D
    fg = GS['fg'][0]
    lights = GS['lights'][fg]                                                        # worldspace_point_lights[fg_mask.squeeze(0)]
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    for tag, lut in (('tm', (GS['lut_x'], GS['lut_y'])), ('plain', (None, None))):
        out = D.shade_points(GS['albedo'], GS['normal'], lights, 0.2, *lut)
        ga, gn = D.shade_points(GS['albedo'], GS['normal'], lights, 0.2, *lut, grad_out=GS['gy'])
        assert rel(out, GS[f'{tag}_mesh_out']) < 3e-6 and rel(ga, GS[f'{tag}_mesh_g_albedo']) < 3e-6 and rel(gn, GS[f'{tag}_mesh_g_normal']) < 3e-6, tag
        # the NeRF variant: same kernel fed with the decoder's colour; the chain into the decoder weights closes through torch
        W = torch.from_numpy(GS['W']).requires_grad_(True)
        col = torch.sigmoid(torch.from_numpy(GS['pos']) @ W)
        out = D.shade_points(col.detach().numpy(), GS['normal'], lights, 0.2, *lut)
        ga, gn = D.shade_points(col.detach().numpy(), GS['normal'], lights, 0.2, *lut, grad_out=GS['gy'])
        gW, = torch.autograd.grad(col, W, torch.from_numpy(ga.astype(np.float64)))
        assert rel(out, GS[f'{tag}_nerf_out']) < 3e-6 and rel(gW.numpy(), GS[f'{tag}_nerf_g_W']) < 3e-6 and rel(gn, GS[f'{tag}_nerf_g_normal']) < 3e-6, tag


@pytest.mark.gpu
def test_hip_shading_funs_vs_reference():
    from mvedit_amd.tonemapping import Tonemapping, make_nerf_albedo_shading_fun, make_nerf_shading_fun, make_shading_fun
    cu = lambda k: torch.from_numpy(GS[k]).float().cuda()
    fg = torch.from_numpy(GS['fg']).cuda()
    rel = lambda a, k: float((a.detach().cpu().double() - torch.from_numpy(GS[k])).abs().max() / max(np.abs(GS[k]).max(), 1e-12))
    for tag, tm in (('tm', Tonemapping(device='cuda')), ('plain', None)):
        albedo, normal = cu('albedo').requires_grad_(True), cu('normal').requires_grad_(True)
        y = make_shading_fun(cu('lights'), 0.2, tm)(world_pos=cu('pos'), albedo=albedo, world_normal=normal, fg_mask=fg)
        ga, gn = torch.autograd.grad((y * cu('gy')).sum(), (albedo, normal))
        assert rel(y, f'{tag}_mesh_out') < 3e-6 and rel(ga, f'{tag}_mesh_g_albedo') < 3e-6 and rel(gn, f'{tag}_mesh_g_normal') < 3e-6, tag
        W = cu('W').requires_grad_(True)
        y = make_nerf_shading_fun(lambda x: torch.sigmoid(x @ W), cu('lights'), 0.2, tm)(world_pos=cu('pos'), albedo=None, world_normal=normal, fg_mask=fg)
        gW, gn = torch.autograd.grad((y * cu('gy')).sum(), (W, normal))
        assert rel(y, f'{tag}_nerf_out') < 3e-6 and rel(gW, f'{tag}_nerf_g_W') < 1e-5 and rel(gn, f'{tag}_nerf_g_normal') < 3e-6, tag
    W = cu('W')
    f3 = make_nerf_albedo_shading_fun(lambda x: torch.sigmoid(x @ W))
    assert rel(f3(world_pos=cu('pos')), 'albedo_fun_out') < 1e-6
    empty = torch.zeros(0, 3, device='cuda')
    assert f3(world_pos=empty) is empty and make_nerf_shading_fun(None, None, 0.2)(world_pos=empty, albedo=None) is empty
