"""UNet executor (csrc/unet.hip behind mvedit_amd.unet.UNet2DConditionEngine) vs the torch fp32 oracle.

not gpu : plan-time logic only (no device memory is touched): analytic FLOPs equal the oracle's count and the
          figure SURVEY.md section 8(d) quotes (0.803 TFLOP per 64x64 SD-1.5 forward), op lists are stable, the
          workspace allocator never hands out overlapping live buffers (plan-time guard inside the builder);
          oracle vs committed golden.
gpu     : end-to-end parity on seeded random weights.

Round 5: the engine's DEFAULT mode carries the residual stream as an unrounded pair (16-bit value + 8-bit low half) and is held to north_star's
1e-3 against fp32 arithmetic at every full-size configuration (benchmark shape, rows of the timed 64-image batch, L = 8192 pairing, Zero123++
tiling: 8.4e-4 .. 9.6e-4 measured); the small synthetic configurations (random unit-gain networks of 2-3 levels) measure 8.6e-4 .. 1.1e-3 and keep
the relative criterion below.

Tolerance, stated once.  north_star: "within 1e-3 rel fp16" against the reference's PyTorch path.  The
reference's fp16 path is PyTorch half: every op accumulates in fp32 and rounds its OUTPUT to fp16.  The oracle
reproduces exactly that when run with q=quantizer(float16) (it rounds at every op boundary).  Two numbers are
checked for every case:
   * vs the fp32 oracle: the engine must be at least as close to fp32 truth as the emulated PyTorch-fp16 path is
     (err_engine <= 1.05 * err_emulated + 1e-4): this is the accuracy criterion.  Measured on MI355X (round 1):
     engine 1.27e-3 / 1.72e-3 / 1.38e-3 vs emulated torch-half 1.30e-3 / 1.76e-3 / 1.43e-3 (rel-L2, TINY / SMALL+
     residuals / SMALL paired) -- i.e. PyTorch's own fp16 path is itself 1.3e-3..1.8e-3 away from fp32 on these
     random unit-gain networks, so "1e-3 of the reference" is attainable per kernel, not for ~100 stacked layers;
   * vs the fp16-emulating oracle: rel-L2 <= 3e-3, max-abs/max-abs <= 6e-3 (bf16: 8x): two fp16 evaluations
     with independent rounding noise of the size above, differing in summation order and in where fused ops
     skip an intermediate rounding, sit ~sqrt(2) x that apart.
Per-kernel 1e-3 bounds are enforced in tests/test_unet_ops.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as U

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'unet_tiny.npz')


def inputs(cfg, B, S, seed=0, ctx_len=77):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg['in_channels'], S, S, generator=g)
    ctx = torch.randn(B, ctx_len, cfg['cross_attention_dim'], generator=g)
    return x, ctx


def residuals(cfg, B, S, seed=3, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    ch = cfg['block_out_channels']
    shapes = [(ch[0], S, S)]
    s = S
    for i, c in enumerate(ch):
        shapes += [(c, s, s)] * cfg['layers_per_block']
        if i + 1 < len(ch):
            s //= 2
            shapes.append((c, s, s))
    down = [scale * torch.randn(B, *sh, generator=g) for sh in shapes]
    mid = scale * torch.randn(B, ch[-1], s, s, generator=g)
    return down, mid


# ------------------------------------------------------------------------------------------------ CPU
def test_plan_flops_match_oracle_and_survey(lib):
    from mvedit_amd.unet import UNet2DConditionEngine
    eng = UNet2DConditionEngine(U.SD15, torch.float16, device='cpu')
    info = eng.plan(1, 64, 64, 77)
    fl = info['flops']
    total = fl['conv3x3'] + fl['linear'] + fl['attention']
    # oracle count for the same topology (run once on a 64x64 latent in test_oracle_* is slow; closed form here)
    assert abs(total / 1e12 - 0.8033) < 2e-3, total             # 0.803 TFLOP / image
    assert abs(fl['attention'] / 1e12 - 0.12605) < 1e-4
    assert info['n_ops'] > 270 and info['workspace_bytes'] < 1 << 30          # (round 6: 48 LayerNorms ride with the GEMMs in front of them: 330 -> 282 ops)
    # cross-image pairing doubles self-attention sequence length: 0.926 TFLOP per image (section 8(d))
    info2 = eng.plan(2, 64, 64, 77, num_cross_attn_imgs=2)
    t2 = sum(info2['flops'][k] for k in ('conv3x3', 'linear', 'attention'))
    assert abs(t2 / 2 / 1e12 - 0.926) < 5e-3, t2
    # the BASELINE workload: V=32 views x CFG = 64 images in ONE plan
    info64 = eng.plan(64, 64, 64, 77)
    t64 = sum(info64['flops'][k] for k in ('conv3x3', 'linear', 'attention'))
    assert abs(t64 / 1e12 - 51.4) < 0.2
    assert info64['workspace_bytes'] < 40 << 30
    ops = eng.op_table()
    assert len(ops) == info64['n_ops'] and ops[0][3] == 'nchw->nhwc' and ops[-1][3] == 'nhwc->nchw'
    assert sum(1 for o in ops if o[3] == 'attention') == 32 and sum(1 for o in ops if o[1] == 'conv3x3') == 2 * 22 + 3 + 3 + 2
    assert any(ph == 1 for ph, *_ in ops) and any(ph == 2 for ph, *_ in ops)


def test_plan_rejects_bad_shapes(lib):
    from mvedit_amd.unet import UNet2DConditionEngine
    eng = UNet2DConditionEngine(U.SD15, torch.float16, device='cpu')
    with pytest.raises(lib.MveError, match='divisible'):
        eng.plan(1, 60, 60, 77)
    with pytest.raises(lib.MveError, match='num_cross_attn_imgs'):
        eng.plan(3, 64, 64, 77, num_cross_attn_imgs=2)
    bad = dict(U.SD15, num_heads=(7, 8, 8, 8))
    with pytest.raises(lib.MveError):
        UNet2DConditionEngine(bad, torch.float16, device='cpu')


def test_oracle_flops_and_param_count():
    shapes = U.param_shapes(U.SD15)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 859_520_964       # SD-1.5 UNet parameter count
    sd = U.make_state_dict(U.TINY, seed=1)
    x, ctx = inputs(U.TINY, 2, 16)
    with torch.no_grad():
        out, fl = U.unet_forward(sd, U.TINY, x, 499, ctx, return_flops=True)
    assert out.shape == (2, 4, 16, 16) and torch.isfinite(out).all()
    # enc + dec == forward (the reference's 2-pass split, diffusers.py:57-164)
    with torch.no_grad():
        emb, res, h, c = U.unet_enc(sd, U.TINY, x, 499, ctx)
        out2, _ = U.unet_dec(sd, U.TINY, emb, res, h, ctx)
    assert torch.equal(out, out2)
    # cross-image pairing only changes attention: with one image per group it is the identity
    with torch.no_grad():
        out3 = U.unet_forward(sd, U.TINY, x, 499, ctx, num_cross_attn_imgs=1)
    assert torch.equal(out, out3)


def _golden_case():
    sd = U.make_state_dict(U.TINY, seed=1234)
    x, ctx = inputs(U.TINY, 2, 16, seed=7)
    down, mid = residuals(U.TINY, 2, 16)
    with torch.no_grad():
        a = U.unet_forward(sd, U.TINY, x, 499, ctx)
        b = U.unet_forward(sd, U.TINY, x, torch.tensor([10.0, 900.0]), ctx, num_cross_attn_imgs=2,
                           down_block_additional_residuals=down, mid_block_additional_residual=mid)
    return dict(plain=a.numpy(), paired_residuals=b.numpy())


def test_oracle_matches_committed_golden():
    assert os.path.exists(GOLDEN), 'golden fixture missing (tests/golden/make_unet_golden.py)'
    gold = np.load(GOLDEN)
    out = _golden_case()
    for k in gold.files:
        np.testing.assert_allclose(out[k], gold[k], rtol=1e-4, atol=1e-5, err_msg=k)


# ------------------------------------------------------------------------------------------------ GPU
def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item(), ((a - b).abs().max() / b.abs().max()).item()


def _parity(cfg, B, S, dtype, seed=0, n_img=1, with_res=False, t=499, ctx_len=77):
    from mvedit_amd.unet import UNet2DConditionEngine
    sd = U.make_state_dict(cfg, seed=1234)
    sd_q = {k: v.to(dtype).float() for k, v in sd.items()}          # both sides see the same (rounded) weights
    x, ctx = inputs(cfg, B, S, seed, ctx_len)
    x, ctx = x.to(dtype).float(), ctx.to(dtype).float()
    down = mid = None
    if with_res:
        down, mid = residuals(cfg, B, S)
        down, mid = [d.to(dtype).float() for d in down], mid.to(dtype).float()
    with torch.no_grad():
        ref32 = U.unet_forward(sd_q, cfg, x, t, ctx, n_img, down, mid)
        ref16 = U.unet_forward(sd_q, cfg, x, t, ctx, n_img, down, mid, q=U.quantizer(dtype))
    eng = UNet2DConditionEngine.from_state_dict(sd_q, cfg, dtype)
    kw = dict(cross_attention_kwargs=dict(num_cross_attn_imgs=n_img) if n_img > 1 else None)
    if with_res:
        kw.update(down_block_additional_residuals=[d.to(dtype).cuda() for d in down],
                  mid_block_additional_residual=mid.to(dtype).cuda())
    out = eng(x.to(dtype).cuda(), t, ctx.to(dtype).cuda(), return_dict=False, **kw)[0]
    assert out.dtype == dtype and out.shape == ref32.shape and torch.isfinite(out).all()
    l2_16, mx_16 = _rel(out, ref16)
    l2_32, mx_32 = _rel(out, ref32)
    emu_l2, emu_mx = _rel(ref16, ref32)
    msg = (f'vs fp16-emulating oracle: l2={l2_16:.2e} max={mx_16:.2e}; vs fp32 oracle: l2={l2_32:.2e} max={mx_32:.2e}; '
           f'emulated torch-half vs fp32: l2={emu_l2:.2e} max={emu_mx:.2e}')
    print(msg)
    tol = 3e-3 if dtype == torch.float16 else 2.4e-2
    assert l2_16 <= tol and mx_16 <= 2 * tol, msg
    assert l2_32 <= 1.05 * emu_l2 + 1e-4, msg
    return eng, out, (x, ctx, down, mid)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_engine_tiny(lib, dtype):
    _parity(U.TINY, 2, 16, dtype)


@pytest.mark.gpu
def test_engine_small_all_head_dims_residuals_and_pairing(lib):
    _parity(U.SMALL, 2, 16, torch.float16, with_res=True)
    _parity(U.SMALL, 4, 16, torch.float16, n_img=2, t=torch.tensor([3.0, 3.0, 950.0, 950.0]))
    _parity(U.SMALL, 2, 24, torch.float16, n_img=2, with_res=True, ctx_len=93)     # 77 text + 16 IP tokens; ragged 24x24


@pytest.mark.gpu
def test_engine_sd15_256px(lib):
    """BASELINE config 1: a single 256x256 (32x32 latent) SD-1.5 UNet forward."""
    eng, out, (x, ctx, _, _) = _parity(U.SD15, 1, 32, torch.float16)
    # determinism: bitwise identical on re-run; batch invariance: row b of a batch == the same item alone
    out2 = eng(x.half().cuda(), 499, ctx.half().cuda())[0]
    assert torch.equal(out, out2)


@pytest.mark.gpu
def test_engine_enc_dec_equals_forward_and_fp32_io(lib):
    from mvedit_amd.unet import UNet2DConditionEngine, unet_enc, unet_dec
    cfg, dtype = U.TINY, torch.float16
    sd = U.make_state_dict(cfg, seed=5)
    eng = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype)
    x, ctx = inputs(cfg, 2, 16, seed=2)
    down, mid = residuals(cfg, 2, 16)
    xg, cg = x.half().cuda(), ctx.half().cuda()
    dg, mg = [d.half().cuda() for d in down], mid.half().cuda()
    full = eng(xg, 321, cg, down_block_additional_residuals=dg, mid_block_additional_residual=mg)[0]
    emb, res, h = unet_enc(eng, xg, 321, cg)
    two = unet_dec(eng, emb, res, h, cg, down_block_additional_residuals=dg, mid_block_additional_residual=mg)
    assert torch.equal(full, two)
    # the same cached enc state decoded twice (2-pass mode re-uses it) gives the same answer
    assert torch.equal(two, unet_dec(eng, emb, res, h, cg, down_block_additional_residuals=dg, mid_block_additional_residual=mg))
    plain = eng(xg, 321, cg)[0]
    assert torch.equal(plain, unet_dec(eng, emb, res, h, cg))
    # fp32 I/O at the seam (the reference keeps latents in the runner dtype; fp32 callers must work too)
    out32 = eng(x.cuda(), 321, ctx.cuda())[0]
    assert out32.dtype == torch.float32
    l2, _ = _rel(out32, plain)
    assert l2 < 1e-3


@pytest.mark.gpu
def test_engine_fails_loudly_without_weights(lib):
    from mvedit_amd.unet import UNet2DConditionEngine
    eng = UNet2DConditionEngine(U.TINY, torch.float16)
    x, ctx = inputs(U.TINY, 1, 16)
    sd = U.make_state_dict(U.TINY)
    sd.pop('mid_block.resnets.0.conv1.weight')
    with pytest.raises(KeyError, match='mid_block.resnets.0.conv1.weight'):
        eng.load_state_dict(sd)
    with pytest.raises(lib.MveError, match='not loaded'):
        eng(x.half().cuda(), 1, ctx.half().cuda())


# ------------------------------------------------------------------------------------------------ attention processors
def _check(out, ref16, ref32, dtype=torch.float16):
    l2_16, mx_16 = _rel(out, ref16)
    l2_32, _ = _rel(out, ref32)
    emu_l2, _ = _rel(ref16, ref32)
    msg = f'vs fp16 oracle l2={l2_16:.2e} max={mx_16:.2e}; vs fp32 l2={l2_32:.2e}; emulated l2={emu_l2:.2e}'
    print(msg)
    tol = 3e-3 if dtype == torch.float16 else 2.4e-2                # the bounds of _parity
    assert l2_16 <= tol and mx_16 <= 2 * tol, msg
    assert l2_32 <= 1.05 * emu_l2 + 1e-4, msg


@pytest.mark.gpu
@pytest.mark.parametrize('n_img', [1, 2])
def test_engine_ip_adapter(lib, n_img):
    """IPAttnProcessor2_0 (attention_processor.py:301-396): the last 16 context rows use to_k_ip/to_v_ip in a second
    softmax, added with `scale`; also under the cross-image wrapper (the two compose in the reference)."""
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype, B, S = U.SMALL, torch.float16, 2, 32
    sd = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=8).items()}
    sd.update({k: v.to(dtype).float() for k, v in U.make_ip_state_dict(cfg).items()})
    x, ctx = inputs(cfg, B, S, seed=4, ctx_len=77 + 16)
    x, ctx = x.to(dtype).float(), ctx.to(dtype).float()
    ao = dict(ip_tokens=16, ip_scale=0.6)
    with torch.no_grad():
        ref32 = U.unet_forward(sd, cfg, x, 250, ctx, n_img, attn_opts=ao)
        ref16 = U.unet_forward(sd, cfg, x, 250, ctx, n_img, attn_opts=ao, q=U.quantizer(dtype))
        plain = U.unet_forward(sd, cfg, x, 250, ctx[:, :77], n_img)
    assert _rel(ref32, plain)[0] > 0.02, 'the ip branch must matter in this test'
    eng = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype)
    eng.set_ip_adapter(16, 0.6)
    cak = dict(num_cross_attn_imgs=n_img) if n_img > 1 else None
    out = eng(x.to(dtype).cuda(), 250, ctx.to(dtype).cuda(), cross_attention_kwargs=cak)[0]
    _check(out, ref16, ref32)
    eng.set_ip_adapter(0)                                  # removing the adapter restores the plain processor
    out0 = eng(x.to(dtype).cuda(), 250, ctx[:, :77].to(dtype).cuda(), cross_attention_kwargs=cak)[0]
    assert _rel(out0, plain)[0] < 3e-3
    # without the ip weights the engine refuses
    eng2 = UNet2DConditionEngine.from_state_dict({k: v for k, v in sd.items() if 'processor' not in k}, cfg, dtype)
    eng2.set_ip_adapter(16, 1.0)
    from mvedit_amd._lib import MveError
    with pytest.raises(MveError, match='to_k_ip'):
        eng2(x.to(dtype).cuda(), 250, ctx.to(dtype).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize('cfg_first,S_ref', [(False, 32), (True, 16)])
def test_engine_reference_attention(lib, cfg_first, S_ref):
    """ReferenceAttnProc 'w' then 'r' (diffusers.py:646-673); with is_cfg_guidance the first item neither writes nor reads
    (zero123plus.py:59-76) and the condition latent may have another size than the sample."""
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype, B, S = U.SMALL, torch.float16, 3, 32
    sd = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=9).items()}
    x, ctx = inputs(cfg, B, S, seed=6)
    xr, _ = inputs(cfg, B, S_ref, seed=7)
    x, xr, ctx = x.to(dtype).float(), xr.to(dtype).float(), ctx.to(dtype).float()
    skip = 1 if cfg_first else 0

    def oracle(q):
        d = {}
        U.unet_forward(sd, cfg, xr, 300, ctx, attn_opts=dict(mode='w', ref_dict=d, ref_skip=skip), q=q)
        assert len(d) > 0
        out = U.unet_forward(sd, cfg, x, 300, ctx, attn_opts=dict(mode='r', ref_dict=d, ref_skip=skip), q=q)
        assert len(d) == 0
        return out
    with torch.no_grad():
        ref32, ref16 = oracle(None), oracle(U.quantizer(dtype))
        plain = U.unet_forward(sd, cfg, x, 300, ctx)
    assert _rel(ref32, plain)[0] > 0.02
    eng = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype)
    d = {}
    kw = dict(is_cfg_guidance=True) if cfg_first else {}
    eng(xr.to(dtype).cuda(), 300, ctx.to(dtype).cuda(), cross_attention_kwargs=dict(mode='w', ref_dict=d, **kw))
    assert len(d) == 1
    keep = dict(d)
    out_m = eng(x.to(dtype).cuda(), 300, ctx.to(dtype).cuda(), cross_attention_kwargs=dict(mode='m', ref_dict=d, **kw))[0]
    assert len(d) == 1
    out = eng(x.to(dtype).cuda(), 300, ctx.to(dtype).cuda(), cross_attention_kwargs=dict(mode='r', ref_dict=d, **kw))[0]
    assert len(d) == 0 and torch.equal(out, out_m)
    _check(out, ref16, ref32)
    if cfg_first:   # the unconditional item never sees the reference: equals the plain forward of that item
        p0 = eng(x.to(dtype).cuda(), 300, ctx.to(dtype).cuda())[0]
        assert torch.equal(out[:1], p0[:1]) and not torch.equal(out[1:], p0[1:])
    del keep


@pytest.mark.gpu
def test_engine_sd21_topology_zero123pp_tiling(lib):
    """BASELINE config 2 (Zero123++): SD-2.1 topology -- head_dim 64, use_linear_projection, context dim 1024 -- on the
    reference-true 3:2 latent tiling (120 x 80 scaled down to 48 x 32), with reference-only attention and CFG-first item
    (lib/pipelines/zero123plus.py:107-150)."""
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg = dict(U.SD21, block_out_channels=(320, 640, 1280), layers_per_block=1, down_attn=(True, True, True), num_heads=(5, 10, 20),
               transformer_layers=(1, 1, 1))
    dtype, B = torch.float16, 2
    sd = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=12).items()}
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 4, 48, 32, generator=g).to(dtype).float()          # 3:2 tiling of six views
    cond = torch.randn(B, 4, 16, 16, generator=g).to(dtype).float()       # condition latent (size from the checkpoint's preprocessor)
    ctx = torch.randn(B, 77, 1024, generator=g).to(dtype).float()

    def oracle(q):
        d = {}
        U.unet_forward(sd, cfg, cond, 400, ctx, attn_opts=dict(mode='w', ref_dict=d, ref_skip=1), q=q)
        return U.unet_forward(sd, cfg, x, 400, ctx, attn_opts=dict(mode='r', ref_dict=d, ref_skip=1), q=q)
    with torch.no_grad():
        ref32, ref16 = oracle(None), oracle(U.quantizer(dtype))
    eng = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype)
    d = {}
    eng(cond.to(dtype).cuda(), 400, ctx.to(dtype).cuda(), cross_attention_kwargs=dict(mode='w', ref_dict=d, is_cfg_guidance=True))
    out = eng(x.to(dtype).cuda(), 400, ctx.to(dtype).cuda(), cross_attention_kwargs=dict(mode='r', ref_dict=d, is_cfg_guidance=True))[0]
    assert out.shape == (B, 4, 48, 32)
    _check(out, ref16, ref32)


# ------------------------------------------------------------------------------------------------ benchmark shapes vs the oracle
def _parity_hw(cfg, B, H, W, dtype, n_img=1, seed=0, t=499, ctx_len=77, sd_seed=1234, fp32_bar=None):
    """_parity for a rectangular latent: engine vs the fp32 oracle and vs the oracle emulating PyTorch's half path, same bounds; fp32_bar: the
    engine's DEFAULT mode (residual stream as an unrounded pair) additionally within that rel-L2 of fp32 arithmetic (north_star: 1e-3 in fp16)."""
    from mvedit_amd.unet import UNet2DConditionEngine
    sd_q = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=sd_seed).items()}
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg['in_channels'], H, W, generator=g).to(dtype).float()
    ctx = torch.randn(B, ctx_len, cfg['cross_attention_dim'], generator=g).to(dtype).float()
    with torch.no_grad():
        ref32 = U.unet_forward(sd_q, cfg, x, t, ctx, n_img)
        ref16 = U.unet_forward(sd_q, cfg, x, t, ctx, n_img, q=U.quantizer(dtype))
    eng = UNet2DConditionEngine.from_state_dict(sd_q, cfg, dtype)
    cak = dict(num_cross_attn_imgs=n_img) if n_img > 1 else None
    out = eng(x.to(dtype).cuda(), t, ctx.to(dtype).cuda(), cross_attention_kwargs=cak)[0]
    assert out.dtype == dtype and out.shape == ref32.shape and torch.isfinite(out).all()
    _check(out, ref16, ref32, dtype)
    if fp32_bar is not None:
        assert eng.residual_pair and _rel(out, ref32)[0] <= fp32_bar, (_rel(out, ref32)[0], fp32_bar)
    return eng, out


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
def test_engine_sd15_benchmark_shape_vs_oracle(lib, dtype):
    """The shape bench.py times (SD-1.5, 64x64 latents = 512^2 views; adapter3d_mixin.py:68-135 feeds the CFG pair of every view as one batch)
    compared with the oracle at FULL size: a CFG pair (B = 2) in fp16 and in bf16 -- the reference's default dtype.  fp16: the default engine mode
    is held to north_star's 1e-3 against fp32 arithmetic (round 5; the reference's own half path sits 1.45e-3 from it)."""
    _parity_hw(U.SD15, 2, 64, 64, dtype, fp32_bar=1.0e-3 if dtype == torch.float16 else None)


def _pair_vs_plain(cfg, B, H, W, dtype, n_img=1, with_res=False, t=499, seed=0, bar=None):
    """The engine with the residual stream as an unrounded (hi, lo) pair (set_residual_pair) against the fp32 oracle over the same 16-bit weights,
    next to the plain engine: the pair must be closer to fp32, and under `bar` when one is given."""
    from mvedit_amd.unet import UNet2DConditionEngine
    sd_q = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=1234).items()}
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg['in_channels'], H, W, generator=g).to(dtype).float()
    ctx = torch.randn(B, 77, cfg['cross_attention_dim'], generator=g).to(dtype).float()
    down = mid = None
    if with_res:
        down, mid = residuals(cfg, B, H)
        down, mid = [d.to(dtype).float() for d in down], mid.to(dtype).float()
    with torch.no_grad():
        ref32 = U.unet_forward(sd_q, cfg, x, t, ctx, n_img, down, mid)
    eng = UNet2DConditionEngine.from_state_dict(sd_q, cfg, dtype)
    kw = dict(cross_attention_kwargs=dict(num_cross_attn_imgs=n_img) if n_img > 1 else None)
    if with_res:
        kw.update(down_block_additional_residuals=[d.to(dtype).cuda() for d in down], mid_block_additional_residual=mid.to(dtype).cuda())
    args = (x.to(dtype).cuda(), t, ctx.to(dtype).cuda())
    assert eng.residual_pair                                          # the default mode since round 5
    pair = eng(*args, **kw)[0]
    assert eng.set_residual_pair(False) is True and not eng.residual_pair
    plain = eng(*args, **kw)[0]
    assert eng.set_residual_pair(True) is False and eng.residual_pair
    pair2 = eng(*args, **kw)[0]
    assert torch.equal(pair, pair2)                                   # deterministic; the mode is a plan key: switching back restores the bits
    assert eng.set_residual_pair(False) is True
    assert torch.equal(eng(*args, **kw)[0], plain)
    eng.set_residual_pair(True)
    e_plain, e_pair = _rel(plain, ref32)[0], _rel(pair, ref32)[0]
    print(f'{dtype} {H}x{W} B={B}: rel-L2 vs the fp32 oracle: plain {e_plain:.3e}, residual pair {e_pair:.3e}')
    assert torch.isfinite(pair).all() and e_pair < e_plain, (e_pair, e_plain)
    if bar is not None:
        assert e_pair <= bar, (e_pair, bar)
    return eng, e_plain, e_pair


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
def test_engine_residual_pair_small_cases(lib, dtype):
    """residual_pair mode on the small configurations (128-row kernels, split-K reducer, ControlNet residual sums, cross-image pairing)."""
    _pair_vs_plain(U.TINY, 2, 16, 16, dtype)
    _pair_vs_plain(U.SMALL, 2, 16, 16, dtype, with_res=True)
    _pair_vs_plain(U.SMALL, 4, 16, 16, dtype, n_img=2, t=torch.tensor([3.0, 3.0, 950.0, 950.0]))


@pytest.mark.gpu
def test_engine_residual_pair_meets_north_star_at_the_benchmark_shape(lib):
    """north_star: outputs within 1e-3 rel of the reference CPU path in fp16.  One SD-1.5 forward at the benchmark latent size (64 x 64, CFG pair)
    with the residual stream as an unrounded pair: rel-L2 against the fp32 oracle <= 1e-3 (the plain engine measures 1.2-1.3e-3, PyTorch's own half
    path 1.45e-3; tests/rounding_budget_experiment.py predicts 7.0e-4 for the pair).  bf16: the same mode must at least halve the gap."""
    _pair_vs_plain(U.SD15, 2, 64, 64, torch.float16, bar=1.0e-3)
    _, e_plain, e_pair = _pair_vs_plain(U.SD15, 1, 64, 64, torch.bfloat16)
    assert e_pair <= 0.75 * e_plain, (e_pair, e_plain)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16], ids=['fp16', 'bf16'])
def test_engine_sd15_cross_image_pairing_full_size_vs_oracle(lib, dtype):
    """CrossImageAttnProcWrapper at the size the 3D pipelines use it (joint_attn.py:11-37 with num_cross_attn_imgs = 2 on a [2, 4, 128, 64]
    latent, i.e. one 2 x 8192-token self-attention at level 0): full-size oracle comparison."""
    _parity_hw(U.SD15, 2, 128, 64, dtype, n_img=2, t=torch.tensor([20.0, 20.0]), fp32_bar=1.0e-3 if dtype == torch.float16 else None)      # (measured 9.2e-4)


@pytest.mark.gpu
def test_engine_sd21_zero123pp_true_tiling_vs_oracle(lib):
    """BASELINE config 2 at its true size: the full 4-level SD-2.1 topology on the 3 x 2 tiling of six 320^2 views = a 120 x 80 latent
    (lib/pipelines/zero123plus.py:349-350), reference-only attention written by the condition pass and read by the denoising pass with the
    CFG-first item exempt (zero123plus.py:43-77, :107-150)."""
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype, B = U.SD21, torch.float16, 2
    sd = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=12).items()}
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 4, 120, 80, generator=g).to(dtype).float()
    cond = torch.randn(B, 4, 40, 40, generator=g).to(dtype).float()       # the 320^2 condition image's latent
    ctx = torch.randn(B, 77, 1024, generator=g).to(dtype).float()

    def oracle(q):
        d = {}
        U.unet_forward(sd, cfg, cond, 400, ctx, attn_opts=dict(mode='w', ref_dict=d, ref_skip=1), q=q)
        return U.unet_forward(sd, cfg, x, 400, ctx, attn_opts=dict(mode='r', ref_dict=d, ref_skip=1), q=q)
    with torch.no_grad():
        ref32, ref16 = oracle(None), oracle(U.quantizer(dtype))
    eng = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype)
    d = {}
    eng(cond.to(dtype).cuda(), 400, ctx.to(dtype).cuda(), cross_attention_kwargs=dict(mode='w', ref_dict=d, is_cfg_guidance=True))
    out = eng(x.to(dtype).cuda(), 400, ctx.to(dtype).cuda(), cross_attention_kwargs=dict(mode='r', ref_dict=d, is_cfg_guidance=True))[0]
    assert out.shape == (B, 4, 120, 80)
    _check(out, ref16, ref32)
    assert eng.residual_pair and _rel(out, ref32)[0] <= 1.0e-3, _rel(out, ref32)[0]          # north_star's bar in the default mode (measured 9.3e-4)


def test_synthetic_weights_module_matches_oracle_inventory():
    """mvedit_amd.synthetic (what bench.py uses for weights) and the oracle's own parameter inventory must describe the same
    state dict, name for name and value for value."""
    from mvedit_amd import synthetic as S
    for cfg in (U.SD15, U.SD21, U.TINY):
        assert list(S.param_shapes(cfg).items()) == list(U.param_shapes(cfg).items())
        assert S.controlnet_param_shapes(cfg) == U.controlnet_param_shapes(cfg)
    a, b = S.make_state_dict(U.TINY, seed=3), U.make_state_dict(U.TINY, seed=3)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    a, b = S.make_controlnet_state_dict(U.TINY, seed=4), U.make_controlnet_state_dict(U.TINY, seed=4)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.gpu
def test_engine_sd15_full_size_properties(lib):
    """BASELINE configs 3/4 at full size (SD-1.5, 64x64 latents): no oracle run at this size on the GPU box -- size-independent
    properties instead: determinism, batch / partition invariance (a view's result does not depend on which other views share
    the launch -- bitwise, across the big-tile / small-tile / split-K dispatch), and the CFG combination being linear."""
    from mvedit_amd import ops, synthetic
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype = U.SD15, torch.float16
    eng = UNet2DConditionEngine.from_state_dict(synthetic.make_state_dict(cfg, seed=1234, dtype=dtype), cfg, dtype)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(12, 4, 64, 64, generator=g).to(dtype).cuda()
    ctx = torch.randn(12, 77, 768, generator=g).to(dtype).cuda()
    full = eng(x, 499, ctx)[0]
    assert full.shape == (12, 4, 64, 64) and torch.isfinite(full).all() and full.float().std() > 1e-3
    assert torch.equal(full, eng(x, 499, ctx)[0]), 'deterministic'
    for sl in (slice(0, 1), slice(3, 7), slice(8, 12)):          # 1, 4 and 4 images: other tile shapes, same bits
        assert torch.equal(full[sl], eng(x[sl].contiguous(), 499, ctx[sl].contiguous())[0]), sl
    un, tx = full[:6].float(), full[6:].float()
    a, b = ops.cfg_combine(un, tx, 7.0), ops.cfg_combine(un, tx, 3.0)
    mid = ops.cfg_combine(un, tx, 5.0)
    assert (mid - 0.5 * (a + b)).abs().max() <= 1e-5 * (1 + mid.abs().max())


@pytest.mark.gpu
def test_engine_sd15_rows_of_the_timed_64_image_batch_vs_oracle(lib):
    """The batch bench.py times (32 views x CFG = 64 images, adapter3d_mixin.py:68-135) takes launch decisions no small batch takes: at 64 images
    the 32 x 32 / 16 x 16 levels accumulate K in ONE chain and the 8 x 8 level runs a reduced slice count (csrc/gemm.hip: launch_gemm), where a
    B = 2 forward splits K by the rule.  This test compares ROWS OF THAT BATCH -- view 0's unconditional and text rows (0 and 32) and the last row
    -- with fp32 oracle forwards of the same items: the default mode's 1e-3 (north_star) must hold on the decisions the timed batch takes."""
    from mvedit_amd import _lib, synthetic
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype, B = U.SD15, torch.float16, 64
    esk = _lib.raw('mve_gemm_effective_splitk')
    # the decisions that distinguish this batch: level-2 conv (16 x 16, K = 9 * 1280) un-split at 64 images, split at 2; the 8 x 8 level cut to fewer slices
    assert esk(B * 256, 1280, 9 * 1280, 256) == 1 and esk(2 * 256, 1280, 9 * 1280, 256) > 1
    assert 1 < esk(B * 64, 1280, 9 * 1280, 64) < esk(2 * 64, 1280, 9 * 1280, 64)
    sd = synthetic.make_state_dict(cfg, seed=1234, dtype=dtype)
    eng = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype)
    assert eng.residual_pair
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 4, 64, 64, generator=g).to(dtype)
    ctx = torch.randn(B, 77, 768, generator=g).to(dtype)
    out = eng(x.cuda(), 499, ctx.cuda())[0].float().cpu()
    assert torch.isfinite(out).all()
    sd32 = {k: v.float().cpu() for k, v in sd.items()}
    rows = [0, 32, 63]
    with torch.no_grad():
        ref = U.unet_forward(sd32, cfg, x[rows].float(), 499, ctx[rows].float())
    for k, r in enumerate(rows):
        e = ((out[r] - ref[k]).norm() / ref[k].norm()).item()
        print(f'row {r} of the 64-image batch vs the fp32 oracle: rel-L2 {e:.3e}')
        assert e <= 1.0e-3, (r, e)
    # the same rows alone (small-batch decisions): the decisions differ in fp32 summation order only, which moves a fraction of a percent of the
    # 16-bit outputs of a few launches by one step -- after which every rounding downstream decorrelates: the two results differ like two
    # independent 16-bit evaluations (measured 9.4e-4; 1.3e-3 on the 16-bit stream), each inside the bar against fp32
    alone = eng(x[rows].cuda(), 499, ctx[rows].cuda())[0].float().cpu()
    d = ((alone - out[rows]).norm() / out[rows].norm()).item()
    print(f'rows alone vs rows of the batch: rel-L2 {d:.3e}')
    assert d <= 2e-3, d
    for k in range(len(rows)):
        e = ((alone[k] - ref[k]).norm() / ref[k].norm()).item()
        assert e <= 1.0e-3, ('alone', rows[k], e)


@pytest.mark.gpu
def test_engine_graph_replay_equals_eager(lib):
    """mve_unet_graph (opt-in): the forward is captured on the second call with identical plan + tensor addresses and replayed afterwards.
    Replay must be bitwise equal to the eager result, must follow in-place changes of the inputs, and capture must refuse the legacy
    default stream with a clear message."""
    from mvedit_amd import _lib
    eng, out_eager, (x, ctx, down, mid) = _parity(U.TINY, 2, 16, torch.float16)
    t = torch.full((2,), 999.0, device='cuda')
    eager = eng(x.cuda(), t, ctx.cuda())[0].clone()
    xs, cs = x.cuda().clone(), ctx.cuda().clone()
    assert eng.enable_graph(True) is False                                           # returns the previous setting
    with pytest.raises(_lib.MveError, match='default stream'):
        eng(xs, t, cs)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        outs = []
        for _ in range(4):                                                           # eager, capture + launch, replay, replay
            o = eng(xs, t, cs)[0]
            outs.append(o.clone())
            del o                                                                    # the allocator hands the same block back: addresses repeat
        assert all(torch.equal(o, eager) for o in outs)
        xs.mul_(0.5)                                                                 # same addresses, new contents: the replay must see them
        o2 = eng(xs, t, cs)[0].clone()
    side.synchronize()
    eng.enable_graph(False)
    assert torch.equal(o2, eng(xs, t, cs)[0]) and not torch.equal(o2, eager)


def test_oracle_blocks_against_torch_modules():
    """The UNet oracle's block arithmetic is third-party (diffusers 0.27.2, absent): UNPINNED.  As for the VAE, cross-check it against the
    same blocks assembled INDEPENDENTLY from torch.nn modules and F.scaled_dot_product_attention (ResnetBlock2D: GroupNorm-SiLU-conv /
    time projection / GroupNorm-SiLU-conv / 1x1 shortcut; Transformer2DModel with one BasicTransformerBlock: GroupNorm(eps 1e-6), 1x1
    proj_in, LayerNorm + self-attention, LayerNorm + cross-attention, LayerNorm + GEGLU feed-forward, proj_out, residual) loaded with the
    oracle's state-dict names, and the sinusoidal time embedding against its closed form in float64."""
    nn = torch.nn
    cfg = U.TINY
    sd = U.make_state_dict(cfg, seed=5)
    c = U._Ctx(sd, cfg, None, None)
    g = torch.Generator().manual_seed(0)
    # ---- time embedding: [cos(t w_k), sin(t w_k)], w_k = 10000^(-k / half)
    t = torch.tensor([0.0, 3.0, 499.0, 999.0])
    k = torch.arange(160, dtype=torch.float64)
    ang = t.double()[:, None] * torch.pow(torch.tensor(10000.0, dtype=torch.float64), -k / 160)[None]
    assert (U.timestep_embedding(t, 320).double() - torch.cat([ang.cos(), ang.sin()], -1)).abs().max() < 2e-4      # fp32 angles up to 999 rad
    # ---- ResnetBlock2D with channel change (down_blocks.1.resnets.0: 320 -> 640)
    p = 'down_blocks.1.resnets.0'
    cin, cout = sd[p + '.conv1.weight'].shape[1], sd[p + '.conv1.weight'].shape[0]
    mods = dict(norm1=nn.GroupNorm(32, cin, eps=1e-5), conv1=nn.Conv2d(cin, cout, 3, padding=1), time_emb_proj=nn.Linear(1280, cout),
                norm2=nn.GroupNorm(32, cout, eps=1e-5), conv2=nn.Conv2d(cout, cout, 3, padding=1), conv_shortcut=nn.Conv2d(cin, cout, 1))
    for name, m in mods.items():
        m.load_state_dict({kk: sd[f'{p}.{name}.{kk}'] for kk in ('weight', 'bias')})
    x, temb = torch.randn(2, cin, 8, 8, generator=g), torch.randn(2, 1280, generator=g)
    with torch.no_grad():
        h = mods['conv1'](nn.functional.silu(mods['norm1'](x))) + mods['time_emb_proj'](nn.functional.silu(temb))[:, :, None, None]
        ref = mods['conv_shortcut'](x) + mods['conv2'](nn.functional.silu(mods['norm2'](h)))
        assert (U._resnet(c, p, x, temb) - ref).abs().max() < 1e-4 * ref.abs().max()
    # ---- Transformer2DModel (down_blocks.0.attentions.0: 320 channels, 8 heads of 40, text context 768)
    p, C, heads = 'down_blocks.0.attentions.0', 320, 8
    b = p + '.transformer_blocks.0'
    lin = lambda name, bias=True: (lambda v: nn.functional.linear(v, sd[name + '.weight'], sd[name + '.bias'] if bias else None))
    ln = lambda name: (lambda v: nn.functional.layer_norm(v, (C,), sd[name + '.weight'], sd[name + '.bias'], 1e-5))

    def attn(prefix, q_in, kv_in):
        split = lambda v: v.view(v.shape[0], v.shape[1], heads, C // heads).transpose(1, 2)
        o = nn.functional.scaled_dot_product_attention(split(lin(prefix + '.to_q', False)(q_in)), split(lin(prefix + '.to_k', False)(kv_in)),
                                                       split(lin(prefix + '.to_v', False)(kv_in)))
        return lin(prefix + '.to_out.0')(o.transpose(1, 2).reshape(q_in.shape))
    x, ctx = torch.randn(2, C, 8, 8, generator=g), torch.randn(2, 77, 768, generator=g)
    with torch.no_grad():
        gn = nn.GroupNorm(32, C, eps=1e-6)
        gn.load_state_dict({kk: sd[f'{p}.norm.{kk}'] for kk in ('weight', 'bias')})
        hcl = nn.functional.conv2d(gn(x), sd[p + '.proj_in.weight'], sd[p + '.proj_in.bias']).permute(0, 2, 3, 1).reshape(2, 64, C)
        hcl = hcl + attn(b + '.attn1', ln(b + '.norm1')(hcl), ln(b + '.norm1')(hcl))
        hcl = hcl + attn(b + '.attn2', ln(b + '.norm2')(hcl), ctx)
        val, gate = lin(b + '.ff.net.0.proj')(ln(b + '.norm3')(hcl)).chunk(2, dim=-1)
        hcl = hcl + lin(b + '.ff.net.2')(val * nn.functional.gelu(gate))
        ref = nn.functional.conv2d(hcl.reshape(2, 8, 8, C).permute(0, 3, 1, 2), sd[p + '.proj_out.weight'], sd[p + '.proj_out.bias']) + x
        got = U._transformer(c, p, x, ctx, heads, 1, 1)
        assert (got - ref).abs().max() < 2e-4 * ref.abs().max()


@pytest.mark.gpu
def test_plain_stream_mode_is_the_reference_parity_anchor(lib):
    """the engine's default mode departs from the reference's half-precision rounding points on purpose (the residual stream is not
    rounded after every block); `set_residual_pair(False)` / MVE_RESIDUAL_PAIR=0 is the mode that reproduces them.  One test keeps that mode pinned to
    the oracle that emulates PyTorch's half modules op by op (q = quantizer): it must sit closer to that emulation than the default mode does, and
    within the bound the per-kernel tests allow."""
    from mvedit_amd.unet import UNet2DConditionEngine
    cfg, dtype, B, S = U.SMALL if hasattr(U, 'SMALL') else U.TINY, torch.float16, 2, 16
    sd_q = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=1234).items()}
    x, ctx = inputs(cfg, B, S, seed=4)
    x, ctx = x.to(dtype).float(), ctx.to(dtype).float()
    with torch.no_grad():
        ref16 = U.unet_forward(sd_q, cfg, x, 499, ctx, q=U.quantizer(dtype))
        ref32 = U.unet_forward(sd_q, cfg, x, 499, ctx)
    eng = UNet2DConditionEngine.from_state_dict(sd_q, cfg, dtype)
    assert eng.residual_pair
    pair = eng(x.to(dtype).cuda(), 499, ctx.to(dtype).cuda())[0]
    eng.set_residual_pair(False)
    plain = eng(x.to(dtype).cuda(), 499, ctx.to(dtype).cuda())[0]
    d_plain, d_pair = _rel(plain, ref16)[0], _rel(pair, ref16)[0]
    print(f'vs the half-emulating oracle: plain stream {d_plain:.2e}, default (pair) mode {d_pair:.2e};  vs fp32: plain {_rel(plain, ref32)[0]:.2e}, pair {_rel(pair, ref32)[0]:.2e}')
    assert d_plain <= 3e-3
    assert _rel(pair, ref32)[0] <= _rel(plain, ref32)[0] * 1.02 + 1e-5       # what the default mode buys: closer to fp32
