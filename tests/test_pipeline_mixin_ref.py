"""`Adapter3DMixin.get_noise_pred` orchestration (chunk walk, reference pairing, ControlNet residual padding, CFG / adapter-scale
combine) against the output of the REFERENCE's own method executed over the same deterministic stand-in networks
(tests/golden/mixin_ref.npz, written by tests/golden/make_mixin_golden.py from lib/pipelines/adapter3d_mixin.py:68-135).
Host logic only: runs on the CPU with torch stand-ins in place of the engines; the native engines behind the same mirror are covered
by tests/test_pipeline_mixin.py on the GPU."""
import os

import numpy as np
import pytest
import torch

import stubs
from mvedit_amd.pipelines import Adapter3DMixin

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mixin_ref.npz'))


class Pipe(Adapter3DMixin):
    def __init__(self, fuse):
        self.unet, self.controlnet, self.fuse_chunks = stubs.StubUNet(), stubs.StubControlNet(), fuse


@pytest.mark.parametrize('fuse', [False, True])
@pytest.mark.parametrize('name', ['plain', 'no_depth_adapter_scale', 'paired'])
def test_get_noise_pred_equals_reference_output(name, fuse):
    """fuse=False is the reference's chunk walk; fuse=True concatenates the chunks into one UNet batch (the MI355X-first default):
    both must reproduce what the reference's method returned."""
    kw = stubs.cases()[name]
    with torch.no_grad():
        out = Pipe(fuse).get_noise_pred(**kw)
    np.testing.assert_allclose(out.numpy(), G[name], rtol=1e-5, atol=1e-6)
