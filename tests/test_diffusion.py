"""Denoise-step bookkeeping: noise scales and the x0 prediction."""
import os

import numpy as np
import pytest
import torch


def test_noise_scales_match_reference_output():
    """get_noise_scales (lib/core/diffusion.py:4-21) against outputs of the reference's own function, integer and fractional
    timesteps on SD's scaled-linear schedule (tests/golden/make_reference_py_golden.py).  Host arithmetic: bit-exact."""
    from mvedit_amd.pipelines.diffusion import get_noise_scales
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_py.npz'))
    ab = g['ns_alphas_bar']
    for kind in ('int', 'float'):
        a, b = get_noise_scales(ab, torch.from_numpy(g[f'ns_t_{kind}']), 1000)
        assert np.array_equal(a.numpy(), g[f'ns_{kind}_a']) and np.array_equal(b.numpy(), g[f'ns_{kind}_b']), kind
    with pytest.raises(AssertionError):
        get_noise_scales(ab, torch.tensor([999.5]), 1000)


@pytest.mark.gpu
def test_x0_prediction_matches_formula(lib):
    """pred_original_sample (lib/pipelines/mvedit_3d_pipeline.py:1253-1255): fp32 sub / mul / div, no contraction -> bit-exact
    against the same expression evaluated by torch on the host."""
    from mvedit_amd.pipelines.diffusion import get_noise_scales, predict_x0
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_py.npz'))
    a, b = get_noise_scales(g['ns_alphas_bar'], torch.tensor([617.25]), 1000)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(3, 4, 64, 64, generator=gen)
    e = torch.randn(3, 4, 64, 64, generator=gen).half()
    out = predict_x0(x.cuda(), e.cuda(), a.item(), b.item())
    want = ((x - b * e.float()) / a).half()
    assert out.dtype == torch.float16 and torch.equal(out.cpu(), want)
