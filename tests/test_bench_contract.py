"""bench.py's contract pieces that need no GPU: the argument defaults the driver relies on, the synthetic passes of every workload (shapes, CFG layout,
partitioning), and that the oracle is reachable from bench.py's cpu_baseline only (tests/test_abi.py checks the imports; here the function runs on a
tiny network)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_default_arguments_are_the_drivers(monkeypatch):
    import bench
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.dtype, a.workload, a.views) == (1, 5, 2, 'fp16', 'mvedit32', 32)
    assert not a.plain_stream and not a.no_extra and not a.no_cpu_baseline          # the default line is the default engine mode, with its checker


def test_passes_of_every_workload():
    from tools import bench_parts as P
    from mvedit_amd.unet import SD15_CONFIG, SD21_CONFIG
    from mvedit_amd.parallel import partition_views
    V = 32
    for world in (1, 2, 8):
        seen = 0
        for rank in range(world):
            lo, hi = partition_views(V, world, rank)
            passes, forwards, metric, workload, v_loc, lo2, hi2 = P.make_passes('mvedit32', dict(SD15_CONFIG), V, lo, hi, hi - lo, world, 'cpu', torch.float16)
            (x, t, ctx, n_img, kw), = passes
            assert x.shape == (2 * (hi - lo), 4, 64, 64) and ctx.shape == (2 * (hi - lo), 77, 768) and t.shape == (2 * (hi - lo),) and n_img == 1 and kw is None
            assert torch.equal(x[:hi - lo], x[hi - lo:])                                   # [uncond | text] halves carry the same latents
            assert forwards == 2 * (hi - lo) and f'{forwards * world} SD-1.5 UNet forwards' in workload
            assert metric == 'multi-view denoise-steps/sec (32 views, 512^2)'            # BASELINE.json's metric
            seen += hi - lo
        assert seen == V
    passes, forwards, *_ = P.make_passes('use_reference', dict(SD15_CONFIG), V, 0, V, V, 1, 'cpu', torch.float16)
    (x, t, ctx, n_img, kw), = passes
    assert x.shape == (4 * V, 4, 64, 64) and n_img == 2 and kw == dict(num_cross_attn_imgs=2) and forwards == 4 * V
    passes, forwards, *_ = P.make_passes('zero123pp', dict(SD21_CONFIG), V, 0, V, V, 1, 'cpu', torch.float16)
    assert [tuple(p[0].shape) for p in passes] == [(2, 4, 40, 40), (2, 4, 120, 80)] and forwards == 2
    assert passes[0][4]['mode'] == 'w' and passes[1][4]['mode'] == 'r' and passes[0][4]['ref_dict'] is passes[1][4]['ref_dict']


def test_cpu_baseline_takes_rows_of_a_batch():
    """The checker leg: given rows of a batch it runs the fp32 oracle on exactly those items (here: the tiny network, two rows)."""
    import bench
    from oracle import unet_oracle as U
    cfg = U.TINY
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 4, 16, 16, generator=g).half()
    ctx = torch.randn(5, 77, cfg['cross_attention_dim'], generator=g).half()
    rec, (bx, bctx, bout) = bench.cpu_baseline(cfg, torch.float16, (16, 16), x=x[[0, 3]], ctx=ctx[[0, 3]], repeats=1)
    assert bx.shape == (2, 4, 16, 16) and torch.equal(bx, x[[0, 3]].float()) and bout.shape == (2, 4, 16, 16) and torch.isfinite(bout).all()
    assert rec['kind'] == 'port' and rec['cores'] >= 1 and rec['seconds_per_forward'] > 0 and '2 UNet forward(s)' in rec['sample']
    sd = {k: v.half().float() for k, v in U.make_state_dict(cfg, seed=1234).items()}
    with torch.no_grad():
        one = U.unet_forward(sd, cfg, x[3:4].float(), 499, ctx[3:4].float())
    assert torch.allclose(bout[1:2], one, rtol=1e-4, atol=1e-5)                           # a row of the B = 2 oracle forward == that item alone
