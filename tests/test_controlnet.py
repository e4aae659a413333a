"""ControlNet executor (csrc/unet.hip in ControlNet mode, mvedit_amd.controlnet) vs the torch oracle restatement of diffusers'
ControlNetModel / MultiControlNetModel as the reference calls them (lib/pipelines/adapter3d_mixin.py:101-116).
SURVEY section 8(f) rank 2 ("next" row): same kernels as the UNet, parity bar as tests/test_unet.py."""
import pytest
import torch

from oracle import unet_oracle as U
from test_unet import _rel, inputs


def test_plan_flops_sd15_controlnet(lib):
    from mvedit_amd.controlnet import ControlNetEngine
    eng = ControlNetEngine(U.SD15, torch.float16, device='cpu')
    info = eng.plan(1, 64, 64, 77)
    total = sum(info['flops'][k] for k in ('conv3x3', 'linear', 'attention'))
    # one SD-1.5 ControlNet ~ 0.261 TFLOP/image (encoder + mid) + ~0.02 for the conditioning embedding / zero convs
    assert 0.255e12 < total < 0.30e12, total
    shapes, mid = eng.output_shapes(1, 64, 64)
    assert len(shapes) == 12 and shapes[0] == (320, 64, 64) and shapes[-1] == (1280, 8, 8) and mid == (1280, 8, 8)


def _case(cfg, B, S, seed):
    sd = {k: v.half().float() for k, v in U.make_controlnet_state_dict(cfg, seed=seed).items()}
    x, ctx = inputs(cfg, B, S, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    cond = torch.rand(B, 3, 8 * S, 8 * S, generator=g)
    return sd, x.half().float(), ctx.half().float(), cond.half().float()


@pytest.mark.gpu
def test_controlnet_engine_vs_oracle(lib):
    from mvedit_amd.controlnet import ControlNetEngine
    cfg, dtype, B, S = U.TINY, torch.float16, 2, 16
    sd, x, ctx, cond = _case(cfg, B, S, 1)
    with torch.no_grad():
        d32, m32 = U.controlnet_forward(sd, cfg, x, 300, ctx, cond, 0.7)
        d16, m16 = U.controlnet_forward(sd, cfg, x, 300, ctx, cond, 0.7, q=U.quantizer(dtype))
    eng = ControlNetEngine.from_state_dict(sd, cfg, dtype)
    down, mid = eng(x.half().cuda(), 300, ctx.half().cuda(), cond.half().cuda(), conditioning_scale=0.7)
    assert len(down) == len(d32) == 4
    for got, r16, r32 in zip(list(down) + [mid], list(d16) + [m16], list(d32) + [m32]):
        assert got.shape == r32.shape and got.dtype == dtype
        l2_16, mx_16 = _rel(got, r16)
        l2_32, _ = _rel(got, r32)
        emu, _ = _rel(r16, r32)
        assert l2_16 <= 3e-3 and mx_16 <= 6e-3, (l2_16, mx_16)
        assert l2_32 <= 1.05 * emu + 1e-4, (l2_32, emu)
    # unknown parameter names / a UNet state dict are rejected loudly
    from mvedit_amd._lib import MveError
    with pytest.raises((MveError, KeyError)):
        ControlNetEngine.from_state_dict(U.make_state_dict(cfg), cfg, dtype)


@pytest.mark.gpu
def test_multi_controlnet_feeds_unet_zero_copy(lib):
    """Two nets with different scales are summed in place (MultiControlNetModel); the channels-last outputs go into the UNet
    engine without a layout conversion and give the same bits as NCHW copies of them."""
    from mvedit_amd.controlnet import ControlNetEngine, MultiControlNetEngine
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype, B, S = U.TINY, torch.float16, 3, 16
    sd1, x, ctx, cond1 = _case(cfg, B, S, 2)
    sd2, _, _, cond2 = _case(cfg, B, S, 3)
    multi = MultiControlNetEngine([ControlNetEngine.from_state_dict(sd1, cfg, dtype), ControlNetEngine.from_state_dict(sd2, cfg, dtype)])
    xg, cg = x.half().cuda(), ctx.half().cuda()
    down, mid = multi(xg, 450, cg, [cond1.half().cuda(), cond2.half().cuda()], [0.5, 1.25])
    with torch.no_grad():
        q = U.quantizer(dtype)
        da, ma = U.controlnet_forward(sd1, cfg, x, 450, ctx, cond1, 0.5, q=q)
        db, mb = U.controlnet_forward(sd2, cfg, x, 450, ctx, cond2, 1.25, q=q)
    for got, a, b in zip(list(down) + [mid], list(da) + [ma], list(db) + [mb]):
        l2, mx = _rel(got, a + b)
        assert l2 <= 3e-3 and mx <= 8e-3, (l2, mx)
    usd = {k: v.half().float() for k, v in U.make_state_dict(cfg, seed=21).items()}
    unet = UNet2DConditionEngine.from_state_dict(usd, cfg, dtype)
    out_cl = unet(xg, 450, cg, down_block_additional_residuals=down, mid_block_additional_residual=mid)[0]
    out_nchw = unet(xg, 450, cg, down_block_additional_residuals=[d.contiguous() for d in down], mid_block_additional_residual=mid.contiguous())[0]
    assert torch.equal(out_cl, out_nchw)
    with torch.no_grad():
        ref = U.unet_forward(usd, cfg, x, 450, ctx, 1, [a + b for a, b in zip(da, db)], ma + mb, q=q)
    assert _rel(out_cl, ref)[0] <= 3e-3


@pytest.mark.gpu
def test_get_noise_pred_with_native_controlnets(lib):
    """The 1-pass method of the reference (adapter3d_mixin.py:68-135) end to end on native engines, paired latents included."""
    from mvedit_amd.controlnet import ControlNetEngine, MultiControlNetEngine
    from mvedit_amd.pipelines import Adapter3DMixin
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype, V, S = U.TINY, torch.float16, 2, 16

    class Pipe(Adapter3DMixin):
        pass
    p = Pipe()
    usd = {k: v.half().float() for k, v in U.make_state_dict(cfg, seed=21).items()}
    p.unet = UNet2DConditionEngine.from_state_dict(usd, cfg, dtype)
    sds = [_case(cfg, 1, S, s)[0] for s in (5, 6)]
    p.controlnet = MultiControlNetEngine([ControlNetEngine.from_state_dict(sd, cfg, dtype) for sd in sds])
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(2 * V, 4, S, S, generator=g).half()
    emb = torch.randn(2 * V, 77, 768, generator=g).half()
    img = torch.rand(2 * V, 3, 8 * S, 8 * S, generator=g).half()
    dep = torch.rand(2 * V, 3, 8 * S, 8 * S, generator=g).half()
    out = p.get_noise_pred([lat[:V].cuda(), lat[V:].cuda()], [emb[:V].cuda(), emb[V:].cuda()], [img[:V].cuda(), img[V:].cuda()],
                           [dep[:V].cuda(), dep[V:].cuda()], 500, 0.6, 0.4, 4.0)
    with torch.no_grad():
        q = U.quantizer(dtype)
        da, ma = U.controlnet_forward(sds[0], cfg, lat.float(), 500, emb.float(), img.float(), 0.6, q=q)
        db, mb = U.controlnet_forward(sds[1], cfg, lat.float(), 500, emb.float(), dep.float(), 0.4, q=q)
        full = U.unet_forward(usd, cfg, lat.float(), 500, emb.float(), 1, [a + b for a, b in zip(da, db)], ma + mb, q=q)
    ref = 4.0 * full[V:] + (1 - 4.0) * full[:V]
    bound = (4.0 + 3.0) * 3e-3 * max(full[V:].norm(), full[:V].norm()).item()
    assert (out.float().cpu() - ref).norm().item() <= bound
    # paired latents ([b,4,2H,W]): reference rows get zero residuals (adapter3d_mixin.py:110-116); just has to run and be finite
    lat2 = torch.randn(2 * V, 4, 2 * S, S, generator=g).half().cuda()
    out2 = p.get_noise_pred([lat2], [emb.cuda()], [img.cuda()], [dep.cuda()], 500, 0.6, 0.4, 4.0)
    assert out2.shape == (V, 4, S, S) and torch.isfinite(out2).all()


@pytest.mark.gpu
def test_shared_conditioning_images_are_bit_identical_to_repeated_ones(lib):
    """mve_controlnet_set_cond_repeat (ControlNetEngine.run with fewer conditioning images than batch items): under classifier-free guidance both
    halves of the batch carry the same control images (mvedit_3d_pipeline.py:1232, `ctrl_images.split(diff_bs) * 2`); the embedding then runs once.
    Engine level: every output equals the repeated-image call bit for bit.  Mixin level: the reference's list-times-two call gives the same noise
    prediction as explicitly repeated tensors, and a list whose halves are different objects is not shared."""
    from mvedit_amd.controlnet import ControlNetEngine, MultiControlNetEngine
    from mvedit_amd.pipelines import Adapter3DMixin
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype, V, S = U.TINY, torch.float16, 3, 16
    sd, x, ctx, cond = _case(cfg, 2 * V, S, 7)
    eng = ControlNetEngine.from_state_dict(sd, cfg, dtype)
    half = cond[:V].cuda().half()
    rep = torch.cat([half, half], 0)
    d1, m1 = eng(x.cuda().half(), 500, ctx.cuda().half(), rep, 0.7)
    d1, m1 = [t.clone() for t in d1], m1.clone()
    d2, m2 = eng(x.cuda().half(), 500, ctx.cuda().half(), half, 0.7)
    assert torch.equal(m1, m2) and all(torch.equal(a, b) for a, b in zip(d1, d2))
    assert lib.raw('mve_controlnet_set_cond_repeat')(eng._h, 0) == 2          # the plan option the last call left
    d3, m3 = eng(x.cuda().half(), 500, ctx.cuda().half(), rep, 0.7)           # and back
    assert torch.equal(m1, m3) and lib.raw('mve_controlnet_set_cond_repeat')(eng._h, 0) == 1
    with pytest.raises(AssertionError):
        eng(x.cuda().half(), 500, ctx.cuda().half(), cond[:4].cuda().half(), 0.7)      # 4 images for 6 items

    class Pipe(Adapter3DMixin):
        pass
    p = Pipe()
    p.unet = UNet2DConditionEngine.from_state_dict({k: v.half().float() for k, v in U.make_state_dict(cfg, seed=21).items()}, cfg, dtype)
    p.controlnet = MultiControlNetEngine([eng, ControlNetEngine.from_state_dict(_case(cfg, 1, S, 8)[0], cfg, dtype)])
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(2 * V, 4, S, S, generator=g).half().cuda()
    emb = torch.randn(2 * V, 77, 768, generator=g).half().cuda()
    dep = torch.rand(V, 3, 8 * S, 8 * S, generator=g).half().cuda()
    chunks = lambda t: list(t.split(2))
    shared = p.get_noise_pred(chunks(lat), chunks(emb), chunks(half) * 2, chunks(dep) * 2, 500, 0.6, 0.4, 4.0)
    calls = []
    run0 = ControlNetEngine.run
    try:
        ControlNetEngine.run = lambda self, s_, t_, e_, c_, *a, **k: (calls.append(c_.shape[0]), run0(self, s_, t_, e_, c_, *a, **k))[1]
        explicit = p.get_noise_pred(chunks(lat), chunks(emb), chunks(half) + chunks(half.clone()), chunks(dep) + chunks(dep.clone()), 500, 0.6, 0.4, 4.0)
        assert calls == [2 * V, 2 * V]                                         # different objects: nothing is assumed about their contents
        calls.clear()
        p.get_noise_pred(chunks(lat), chunks(emb), chunks(half) * 2, chunks(dep) * 2, 500, 0.6, 0.4, 4.0)
        assert calls == [V, V]
        # round 5: the form of the reference's ordinary branch -- `torch.cat([x] * 2).split(diff_bs)`, views of one tensor with equal halves -- is
        # recognised too (Adapter3DMixin._cat_shared_cond: one device comparison of the halves)
        calls.clear()
        cat_split = p.get_noise_pred(chunks(lat), chunks(emb), chunks(torch.cat([half] * 2)), chunks(torch.cat([dep] * 2)), 500, 0.6, 0.4, 4.0)
        assert calls == [V, V]
    finally:
        ControlNetEngine.run = run0
    assert torch.equal(shared, explicit) and torch.equal(shared, cat_split)
    # the 2-pass methods (adapter3d_mixin.py:137-317) on the same engines: depth net in pass 1, tile + depth in pass 2 (restored in round 5)
    two_pass = []
    for imgs, deps in ((chunks(half) * 2, chunks(dep) * 2), (chunks(half) + chunks(half.clone()), chunks(dep) + chunks(dep.clone()))):
        n1, dec_args, dec_kwargs = p.get_noise_pred_p1(chunks(lat), chunks(emb), 500, 4.0, ctrl_depths_batches=deps, depth_weight=0.4)
        two_pass.append((n1, p.get_noise_pred_p2(chunks(lat), chunks(emb), dec_args, dec_kwargs, 500, 4.0, imgs, 0.6, ctrl_depths_batches=deps,
                                                 depth_weight=0.4)))
    assert torch.equal(two_pass[0][0], two_pass[1][0]) and torch.equal(two_pass[0][1], two_pass[1][1])
