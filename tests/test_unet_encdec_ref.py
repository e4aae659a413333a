"""oracle/unet_oracle.py: unet_enc / unet_dec against the reference's in-tree functions of the same names
(lib/models/architecture/diffusers.py:57-164) EXECUTED over a stand-in UNet assembled from the oracle's own block functions
(tests/golden/unet_encdec_ref.npz, written by tests/golden/make_unet_encdec_golden.py): skip bookkeeping, ControlNet residual
additions, per-block skip slicing, the (emb, skips, sample) hand-over between the halves, cross-image attention passed through.
The blocks themselves (diffusers) stay unpinned -- this pins how the reference's own code walks them."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as U

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, 'golden', 'unet_encdec_ref.npz'))
spec = importlib.util.spec_from_file_location('make_unet_encdec_golden', os.path.join(HERE, 'golden', 'make_unet_encdec_golden.py'))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)


@pytest.mark.parametrize('tag,n_img,with_res', [('plain', 1, False), ('controlnet', 1, True), ('paired', 2, True)])
def test_enc_dec_equal_reference_functions(tag, n_img, with_res):
    cfg, sd, x, ctx, down, mid = M.case()
    with torch.no_grad():
        emb, skips, sample, c = U.unet_enc(sd, cfg, x, 321, ctx, n_img)
        out, _ = U.unet_dec(sd, cfg, emb, skips, sample, ctx, n_img, down if with_res else None, mid if with_res else None, _c=c)
    assert np.array_equal(np.array([list(s.shape) for s in skips]), G[f'{tag}_skip_shapes'])
    mom = np.array([[float(s.double().mean()), float(s.double().abs().mean())] for s in skips])
    np.testing.assert_allclose(mom, G[f'{tag}_skip_moments'], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(emb.numpy(), G[f'{tag}_emb'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sample.numpy(), G[f'{tag}_mid'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out.numpy(), G[f'{tag}_out'], rtol=1e-5, atol=1e-6)
