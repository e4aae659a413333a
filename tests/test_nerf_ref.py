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
