"""End-to-end parity of the engine's DEFAULT mode against fp32 oracle arithmetic, on the cases round 5 only ran as experiments
(tests/parity_sweep_experiment.py) or not at all ("Turn the parity evidence into collected tests"):

  * a sweep over input seeds and timesteps at the benchmark latent size (north_star: 1e-3 rel in fp16; the worst case is printed);
  * rows of the TIMED 64-image batch with NONZERO ControlNet residuals (adapter3d_mixin.py:101-125: the reference adds the MultiControlNet
    residuals to every skip and to the mid block -- bench.py's headline runs with zero residuals);
  * rows of a `use_reference` batch (CrossImageAttnProcWrapper, joint_attn.py:11-37: pairs of views attend jointly, 8192 tokens at level 0).

Scope of the claim (stated wherever 1e-3 is quoted): seeded random weights of the SD-1.5 topology, N(0, 1) inputs.  A real checkpoint's activation
statistics (outlier channels) are untested by construction -- there is no checkpoint in the image; fp16 is the only dtype with a tolerance
claim (bf16 measures 7e-3: tests/test_unet.py).  The oracle forwards run on the host cores (a few seconds each)."""
import os

import pytest
import torch

from oracle import unet_oracle as U
from test_unet import residuals

pytestmark = pytest.mark.gpu
BAR = 1.0e-3


def _setup():
    from mvedit_amd import synthetic
    from mvedit_amd.unet import UNet2DConditionEngine
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    cfg, dtype = U.SD15, torch.float16
    sd = synthetic.make_state_dict(cfg, seed=1234, dtype=dtype)
    eng = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype)
    assert eng.residual_pair, 'the default mode is the residual stream as an unrounded pair'
    return cfg, dtype, eng, {k: v.float().cpu() for k, v in sd.items()}


def _rel(a, b):
    return float((a.float().cpu() - b).norm() / b.norm())


def test_default_mode_seed_and_timestep_sweep(lib):
    """4 input seeds x 3 timesteps, one SD-1.5 forward each at 64 x 64 latents: every case inside 1e-3 of fp32 (round 5's experiment script measured
    20 / 20 inside with a worst case of 9.9e-4: the margin is 1-10 %, and this test is what watches it)."""
    cfg, dtype, eng, sd32 = _setup()
    worst = 0.0
    for seed in range(4):
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(1, 4, 64, 64, generator=g).to(dtype)
        ctx = torch.randn(1, 77, 768, generator=g).to(dtype)
        for t in (981, 499, 21):
            with torch.no_grad():
                ref = U.unet_forward(sd32, cfg, x.float(), t, ctx.float())
            e = _rel(eng(x.cuda(), t, ctx.cuda())[0], ref)
            worst = max(worst, e)
            print(f'seed {seed} t {t:4d}: rel-L2 vs the fp32 oracle {e:.3e}', flush=True)
            assert e <= BAR, (seed, t, e)
    print(f'worst of 12: {worst:.3e} (bar {BAR:.0e})')


def test_rows_of_the_64_image_batch_with_nonzero_controlnet_residuals(lib):
    """The timed batch (B = 64) with the 12 + 1 ControlNet residual tensors NONZERO (N(0, 0.3^2), NCHW as the reference passes them): rows 0, 32 and
    63 against fp32 oracle forwards of the same items with the same residual rows."""
    cfg, dtype, eng, sd32 = _setup()
    B = 64
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 4, 64, 64, generator=g).to(dtype)
    ctx = torch.randn(B, 77, 768, generator=g).to(dtype)
    down, mid = residuals(cfg, B, 64, seed=3, scale=0.3)
    down, mid = [d.to(dtype) for d in down], mid.to(dtype)
    out = eng(x.cuda(), 499, ctx.cuda(), down_block_additional_residuals=[d.cuda() for d in down], mid_block_additional_residual=mid.cuda())[0].float().cpu()
    assert torch.isfinite(out).all()
    rows = [0, 32, 63]
    with torch.no_grad():
        ref = U.unet_forward(sd32, cfg, x[rows].float(), 499, ctx[rows].float(), 1, [d[rows].float() for d in down], mid[rows].float())
    for k, r in enumerate(rows):
        e = float((out[r] - ref[k]).norm() / ref[k].norm())
        print(f'row {r} of the 64-image batch, nonzero ControlNet residuals: rel-L2 vs the fp32 oracle {e:.3e}')
        assert e <= BAR, (r, e)
    # and the residuals matter: the same rows without them differ by far more than the bar
    plain = eng(x[rows].cuda(), 499, ctx[rows].cuda())[0].float().cpu()
    assert float((plain - out[rows]).norm() / out[rows].norm()) > 20 * BAR


def test_rows_of_a_use_reference_batch(lib):
    """`use_reference` (num_cross_attn_imgs = 2): a batch of 16 pairs of views; the first and the last pair against fp32 oracle forwards of those pairs
    (the level-0 self-attention of a pair is one 8192-token sequence)."""
    cfg, dtype, eng, sd32 = _setup()
    B = 32
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 4, 64, 64, generator=g).to(dtype)
    ctx = torch.randn(B, 77, 768, generator=g).to(dtype)
    t = torch.full((B,), 261.0)
    out = eng(x.cuda(), t.cuda(), ctx.cuda(), cross_attention_kwargs=dict(num_cross_attn_imgs=2))[0].float().cpu()
    for pair in (0, B // 2 - 1):
        rows = [2 * pair, 2 * pair + 1]
        with torch.no_grad():
            ref = U.unet_forward(sd32, cfg, x[rows].float(), t[rows], ctx[rows].float(), 2)
        e = float((out[rows] - ref).norm() / ref.norm())
        print(f'pair {pair} of the use_reference batch: rel-L2 vs the fp32 oracle {e:.3e}')
        assert e <= BAR, (pair, e)
