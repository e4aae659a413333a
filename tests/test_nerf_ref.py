"""The oracle's restatement of the NeRF inference loop (oracle/nerf_oracle.py: render_rays_eval) against the reference's own
`VolumeRenderer.forward` EXECUTED over the same operator stand-ins (tests/golden/volume_renderer_ref.npz, written by
tests/golden/make_volume_renderer_golden.py from lib/models/decoders/base_volume_renderer.py:264-329).  The operators inside are the
C oracle's, themselves pinned against the reference's kernels (tests/test_raymarching_ref.py): together this pins the oracle that the
fused HIP renderer is tested against (tests/test_nerf.py) down to the hash-grid decode, which stays unpinned (tiny-cuda-nn absent)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import nerf_oracle as N

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, 'golden', 'volume_renderer_ref.npz'))


def _scene():
    spec = importlib.util.spec_from_file_location('make_vr_golden', os.path.join(HERE, 'golden', 'make_volume_renderer_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.scene()


@pytest.mark.parametrize('tag,dt_gamma', [('plain', 0.0), ('dt_gamma', 1.0 / 64)])
def test_eval_loop_restatement_equals_reference_output(tag, dt_gamma):
    params, bits, grid, rays_o, rays_d = _scene()
    ws, depth, image, n_samples = N.render_rays_eval(rays_o, rays_d, bits, grid, params, bound=1.0, min_near=0.2, dt_gamma=dt_gamma, max_steps=256)
    assert n_samples > 1000 and (ws > 0.5).sum() > 20                  # the sphere is hit and composited
    assert np.array_equal(ws, G[f'{tag}_weights_sum']) and np.array_equal(depth, G[f'{tag}_depth']) and np.array_equal(image, G[f'{tag}_image'])


def test_train_branch_restatement_equals_reference_output():
    """oracle/nerf_oracle.py: train_forward (two-pass march, culling by composited weight with the re-indexed ray table, decode,
    compositing) against the reference's forward in training mode, bit for bit including the culled sample list."""
    params, bits, grid, rays_o, rays_d = _scene()
    r = N.train_forward(rays_o, rays_d, bits, grid, params, np.zeros(rays_o.shape[0], np.float32), dt_gamma=0.0, bound=1.0, min_near=0.2,
                        max_steps=256, weight_culling_th=1e-3)
    assert r['weights'].shape[0] > 500 and r['weights'].shape[0] == G['train_weights'].shape[0]
    for k in ('weights', 'weights_sum', 'depth', 'image', 'rays', 'ts'):
        assert np.array_equal(r[k], G[f'train_{k}']), k


def test_nerf_render_restatement_equals_reference_output():
    """oracle/nerf_oracle.py: nerf_render (rays from intrinsics / poses, dt_gamma from the focal lengths, 1/r -> 1/z depth, normals from
    the foreground depth, background blend) against the reference's `BaseNeRF.render` executed over its own geometry helpers and the
    forward above."""
    params, bits, grid, _, _ = _scene()
    rgba, depth, normal, normal_fg = N.nerf_render(params, bits, grid, 16, 16, G['render_intrinsics'], G['render_poses'], dt_gamma_scale=0.5,
                                                   max_steps=256)
    assert (rgba[..., 3] > 0.5).sum() > 30
    np.testing.assert_allclose(rgba, G['render_rgba'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(depth, G['render_depth'], rtol=1e-5, atol=1e-6)
    # normal_fg divides the depth by max(alpha, 1e-6): where alpha ~ 0 it amplifies last-place differences of the depth a million times and
    # is multiplied by alpha again in `normal`; compare it on the foreground only
    fg = (G['render_rgba'][..., 3] > 1e-3)
    for yy in (-1, 0, 1):          # ... whose 4-neighbourhood (the finite-difference stencil of depth_to_normal) is foreground as well
        for xx in (-1, 0, 1):
            fg &= np.roll(G['render_rgba'][..., 3] > 1e-3, (yy, xx), axis=(1, 2))
    assert fg.sum() > 30
    np.testing.assert_allclose(normal_fg[fg], G['render_normal_fg'][fg], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(normal, G['render_normal'], rtol=1e-4, atol=2e-5)


def test_density_grid_refresh_equals_reference_output():
    """oracle/nerf_oracle.py: density_grid_points / density_grid_update (+ the C oracle's packbits) against the reference's
    `update_extra_state` executed on the CPU, full refresh -- the only branch the pipelines reach: they always pass iter_density = 0
    (mvedit_3d_pipeline.py:496-510), and the partial branch raises a shape error for one scene (:163).  The reference's jitter is
    replayed from torch's CPU generator (torch.rand_like over the ij-meshgrid order)."""
    import importlib.util
    import torch
    from oracle import raymarching as ORM
    spec = importlib.util.spec_from_file_location('make_vr_golden', os.path.join(HERE, 'golden', 'make_volume_renderer_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    H, n_cells = 16, 16 ** 3
    dens = lambda x: mod.density_fn(torch.from_numpy(x)).numpy()
    # full refresh: every cell once, jitter = torch.rand_like(xyzs) over the ij-meshgrid order
    torch.manual_seed(11)
    g = np.arange(H)
    coords = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3)
    noise = torch.rand(n_cells, 3).numpy()
    xyzs, idx = N.density_grid_points(coords, noise, H, 1.0)
    grid, mean = N.density_grid_update(np.zeros(n_cells, np.float32), dens(xyzs), idx, 0.9)
    assert np.array_equal(grid, G['grid_full_after'][0]) and (grid > 0.01).sum() > 100
    assert np.array_equal(ORM.packbits(grid, min(float(mean), 0.01)), G['grid_full_bits'][0])


def test_point_decode_composition_equals_reference_output():
    """oracle/nerf_oracle.py: point_decode against the reference's `iNGPDecoder.point_decode` EXECUTED around the oracle's own hash-grid
    encoder (tests/golden/decoder_ref.npz, tests/golden/make_decoder_golden.py): coordinate normalisation, MLP, density blob, truncated-exp
    density, saturated sigmoid.  Also the clamped gradient of the reference's `_trunc_exp` and the constructor's level-scale expression."""
    import importlib.util
    import torch
    D = np.load(os.path.join(HERE, 'golden', 'decoder_ref.npz'))
    spec = importlib.util.spec_from_file_location('make_decoder_golden', os.path.join(HERE, 'golden', 'make_decoder_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    params = N.make_nerf_params(seed=7, table_scale=0.5)
    params['b1'], params['b2'] = D['b1'], D['b2']
    sig, rgb = N.point_decode(mod.points().numpy(), params, 1.0)
    np.testing.assert_allclose(sig, D['sigmas'], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(rgb, D['rgbs'], rtol=1e-5, atol=1e-6)
    assert sig.std() > 0.05 and rgb.std() > 0.01 and (sig[:200] > sig[200:].mean()).mean() > 0.5     # non-degenerate; the blob raises the centre
    # the level scale every hash-grid level derives from
    pls = np.exp2(np.log2(320 * 1.0 / 16) / 11)
    assert pls == float(D['per_level_scale'])
    meta, _ = N.grid_meta(12, 16, 320, 1.0)
    assert abs(float(meta[11][0]) - (16 * pls ** 11 - 1)) < 1e-3 and meta[0][1] == 16
    # d sigma / d pre-activation as the training kernels use it: g * clamp(exp(x), 1e-6, 1e6)
    pre = D['trunc_exp_pre']
    np.testing.assert_allclose(np.clip(np.exp(pre.astype(np.float64)), 1e-6, 1e6), D['trunc_exp_grad'], rtol=1e-6)
