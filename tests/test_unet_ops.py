"""UNet primitive kernels (C ABI section 2) against plain torch fp32 on CPU -- the same primitives
(F.linear / F.conv2d / F.group_norm / F.layer_norm / softmax attention) the UNet oracle is built from.

Tolerance (stated once): inputs are rounded to the storage dtype first and fed identically to both
sides; the HIP kernels accumulate in fp32 and round the result once to the storage dtype, so
    max|hip - ref| / max|ref|  <= 1e-3 (fp16)   /  8e-3 (bf16: 2^-8 output rounding)
and the relative L2 error is bounded by the same number.
"""
import math

import numpy as np
import pytest
import torch

# mve_gemm_tune settings under which every GEMM / conv result must be bit-identical: the 128-row kernel only, the 256-row tile from one
# block up (ping-pong main loop where eligible, csrc/gemm_pp.hip), the same with the ping-pong loop disabled (two-stage loop, gemm_big.hip)
STRICT = 1 << 30          # mve_gemm_tune bit 30: chip-filling launches emulate the K slices (bitwise equal to split-K + reducer) instead of one accumulation chain
TILE_MODES = (0, 1 | STRICT, 1 | (1 << 27) | STRICT)
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def check(name, got, ref, dtype, extra=''):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f'{name}: non-finite output {extra}'
    err = (got - ref).abs()
    rel_max = (err.max() / ref.abs().max().clamp_min(1e-12)).item()
    rel_l2 = ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()
    if not (rel_max <= TOL[dtype] and rel_l2 <= TOL[dtype]):
        idx = np.unravel_index(int(err.argmax()), err.shape)
        bad = (err > 10 * TOL[dtype] * ref.abs().max()).float()
        rows_bad = bad.reshape(bad.shape[0], -1).mean(1)
        msg = (f'{name} {extra}: rel_max={rel_max:.3e} rel_l2={rel_l2:.3e} tol={TOL[dtype]:.0e}; worst at {idx}: '
               f'got {got[idx].item():.5f} ref {ref[idx].item():.5f}; frac bad={bad.mean().item():.4f}; '
               f'first bad rows={torch.nonzero(rows_bad > 0)[:8].flatten().tolist()}')
        raise AssertionError(msg)
    return rel_max, rel_l2


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (256, 320, 320), (1000, 640, 328), (77, 1280, 768), (4096, 320, 2880),
                                   (130, 8, 40), (512, 2560, 320), (64, 1280, 1280), (300, 200, 136)])
def test_gemm_plain(lib, dtype, M, N, K):
    from mvedit_amd import ops
    a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    out = ops.gemm(a.cuda(), w.cuda())
    check('gemm', out, a.float() @ w.float().t(), dtype, f'M={M} N={N} K={K}')


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_gemm_epilogues(lib, dtype):
    from mvedit_amd import ops
    M, N, K = 2 * 333, 640, 320
    a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    bias = rnd((N,), torch.float32, 3)
    rowvec = rnd((2, N), torch.float32, 4)
    res = rnd((M, N), dtype, 5)
    ref = a.float() @ w.float().t() + bias + rowvec.repeat_interleave(333, 0) + res.float()
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), rowvec=rowvec.cuda(), rows_per_vec=333, residual=res.cuda())
    check('gemm+bias+rowvec+residual', out, ref, dtype)
    out32 = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), flags=ops.OUT_F32, out_scale=0.5)
    assert out32.dtype == torch.float32
    check('gemm f32 out', out32, 0.5 * (a.float() @ w.float().t() + bias), dtype)
    # GEGLU: interleaved (value, gate) rows, as the engine packs diffusers' GEGLU.proj
    wv, wg = rnd((N, K), dtype, 6, K ** -0.5), rnd((N, K), dtype, 7, K ** -0.5)
    bv, bg = rnd((N,), torch.float32, 8), rnd((N,), torch.float32, 9)
    w_il = torch.stack([wv, wg], 1).reshape(2 * N, K).contiguous()
    b_il = torch.stack([bv, bg], 1).reshape(2 * N).contiguous()
    ref = (a.float() @ wv.float().t() + bv) * F.gelu(a.float() @ wg.float().t() + bg)
    out = ops.gemm(a.cuda(), w_il.cuda(), bias=b_il.cuda(), flags=ops.GEGLU)
    assert out.shape == (M, N)
    check('gemm geglu', out, ref, dtype)
    # strided A and residual views (a column slice of a wider buffer)
    wide = rnd((M, K + 64), dtype, 10).cuda()
    out = ops.gemm(wide[:, 64:], w.cuda())
    check('gemm strided A', out, wide[:, 64:].float().cpu() @ w.float().t(), dtype)


def conv_ref(x_nchw, w_oihw, bias=None, stride=1, upsample=False):
    x = x_nchw.float()
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode='nearest')
    return F.conv2d(x, w_oihw.float(), bias, stride=stride, padding=1)


def to_nhwc(x):  # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('B,H,W,C1,C2,Cout,stride,ups', [
    (2, 16, 16, 64, 0, 128, 1, False),
    (1, 64, 64, 320, 0, 320, 1, False),      # the heaviest SD-1.5 shape (one image)
    (3, 8, 8, 8, 0, 320, 1, False),          # conv_in (4 latent channels zero padded to 8)
    (2, 12, 20, 320, 0, 8, 1, False),        # conv_out (4 output channels padded to 8), non-square
    (2, 16, 16, 320, 0, 320, 2, False),      # Downsample2D
    (2, 8, 8, 640, 0, 640, 1, True),         # Upsample2D: nearest 2x fused into the gather
    (2, 8, 8, 640, 320, 320, 1, False),      # skip-concat input (two sources)
    (5, 9, 7, 72, 40, 88, 1, False),         # ragged everything
    (2, 9, 7, 72, 0, 88, 2, False),          # odd size, stride 2
])
def test_conv3x3(lib, dtype, B, H, W, C1, C2, Cout, stride, ups):
    from mvedit_amd import ops
    x1 = rnd((B, C1, H, W), dtype, 1)
    x2 = rnd((B, C2, H, W), dtype, 2) if C2 else None
    C = C1 + C2
    w = rnd((Cout, C, 3, 3), dtype, 3, (9 * C) ** -0.5)
    bias = rnd((Cout,), torch.float32, 4)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    ref = conv_ref(xin, w, bias, stride, ups)
    layouts = [False, True] if (C1 % 64 == 0 and C2 % 64 == 0) else [False]
    for chunk64 in layouts:      # [Cout][kh][kw][Cin]  and  [Cout][Cin/64][kh][kw][64]
        w_k, wflag = ops.pack_conv_weight(w, chunk64)
        out, Ho, Wo = ops.conv3x3(to_nhwc(x1).cuda(), w_k.cuda(), B, H, W, x2=to_nhwc(x2).cuda() if C2 else None,
                                  stride=stride, upsample=ups, bias=bias.cuda(), flags=wflag)
        assert (Ho, Wo) == tuple(ref.shape[2:])
        check('conv3x3', out, to_nhwc(ref), dtype, f'{(B, H, W, C1, C2, Cout, stride, ups)} chunk64={chunk64}')


def test_conv3x3_resblock_epilogue(lib):
    """conv + bias + per-image time embedding + residual, as ResnetBlock2D uses it."""
    from mvedit_amd import ops
    dtype = torch.float16
    B, H, W, C = 3, 16, 16, 320
    x = rnd((B, C, H, W), dtype, 1)
    w = rnd((C, C, 3, 3), dtype, 2, (9 * C) ** -0.5)
    bias, temb = rnd((C,), torch.float32, 3), rnd((B, C), torch.float32, 4)
    res = rnd((B, C, H, W), dtype, 5)
    ref = conv_ref(x, w, bias) + temb[:, :, None, None] + res.float()
    w_k, wflag = ops.pack_conv_weight(w)
    out, _, _ = ops.conv3x3(to_nhwc(x).cuda(), w_k.cuda(), B, H, W, bias=bias.cuda(), rowvec=temb.cuda(),
                            residual=to_nhwc(res).cuda(), flags=wflag)
    check('conv3x3 resblock epilogue', out, to_nhwc(ref), dtype)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('B,HW,C1,C2,silu,eps', [(2, 64 * 64, 320, 0, True, 1e-5), (3, 16 * 16, 1280, 640, True, 1e-5),
                                                  (2, 8 * 8, 2560, 0, True, 1e-5), (2, 32 * 32, 640, 0, False, 1e-6),
                                                  (1, 33, 640, 320, True, 1e-5),
                                                  # one-launch path on 1024 threads, XCD-grouped block order (B % 8 == 0)
                                                  (8, 32 * 32, 640, 320, True, 1e-5), (8, 32 * 32, 1280, 1280, True, 1e-5),
                                                  # the VAE's narrow tensors (16- and 32-lane layouts, many row splits)
                                                  (2, 96 * 96, 128, 0, True, 1e-6), (1, 64 * 64, 256, 0, False, 1e-6),
                                                  (2, 700, 64, 64, True, 1e-6), (3, 231, 32, 0, True, 1e-6)])
def test_groupnorm(lib, dtype, B, HW, C1, C2, silu, eps):
    from mvedit_amd import ops
    C = C1 + C2
    x1 = rnd((B * HW, C1), dtype, 1) + 0.5
    x2 = (rnd((B * HW, C2), dtype, 2) * 2 - 1).to(dtype) if C2 else None
    gamma, beta = 1 + 0.2 * rnd((C,), torch.float32, 3), 0.2 * rnd((C,), torch.float32, 4)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    xr = xin.float().reshape(B, HW, C).permute(0, 2, 1)
    ref = F.group_norm(xr, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(B * HW, C)
    out = ops.groupnorm(x1.cuda(), B, HW, gamma.cuda(), beta.cuda(), 32, eps, silu, x2.cuda() if C2 else None)
    check('groupnorm', out, ref, dtype, f'{(B, HW, C1, C2, silu)}')


@pytest.mark.parametrize('HW,C1,C2,fused_max_hw', [(32 * 32, 640, 0, 1024), (16 * 16, 1280, 640, 1024), (64 * 64, 320, 0, 1024),
                                                     (64 * 64, 320, 320, 1024), (64 * 64, 320, 0, 4096), (64 * 64, 320, 320, 4096)])
def test_groupnorm_batch_invariance_and_paths(lib, HW, C1, C2, fused_max_hw):
    """An image's GroupNorm output must not depend on how many images share the launch (bitwise), on either path; and the one-launch
    path (statistics out of registers) agrees with the three-launch path to rounding."""
    from mvedit_amd import ops, _lib
    dtype, B = torch.float16, 16
    C = C1 + C2
    x1 = (rnd((B * HW, C1), dtype, 1) + 0.5).cuda()
    x2 = (rnd((B * HW, C2), dtype, 2) * 2 - 1).to(dtype).cuda() if C2 else None
    gamma, beta = (1 + 0.2 * rnd((C,), torch.float32, 3)).cuda(), (0.2 * rnd((C,), torch.float32, 4)).cuda()
    prev = _lib.raw('mve_groupnorm_tune')(fused_max_hw)
    try:
        full = ops.groupnorm(x1, B, HW, gamma, beta, 32, 1e-5, True, x2)
        for b0, nb in ((0, 1), (5, 3), (8, 8)):
            sl = slice(b0 * HW, (b0 + nb) * HW)
            part = ops.groupnorm(x1[sl].contiguous(), nb, HW, gamma, beta, 32, 1e-5, True, x2[sl].contiguous() if C2 else None)
            assert torch.equal(part, full[sl]), f'images {b0}..{b0 + nb} differ between a {nb}-image and a {B}-image launch'
        _lib.raw('mve_groupnorm_tune')(0)
        three = ops.groupnorm(x1, B, HW, gamma, beta, 32, 1e-5, True, x2)
    finally:
        _lib.raw('mve_groupnorm_tune')(prev)
    assert (full.float() - three.float()).abs().max().item() <= 4e-3


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('M,C', [(4096, 320), (1023, 640), (70, 1280), (5, 2048)])
def test_layernorm(lib, dtype, M, C):
    from mvedit_amd import ops
    x = rnd((M, C), dtype, 1) * 3 + 1
    gamma, beta = 1 + 0.2 * rnd((C,), torch.float32, 3), 0.2 * rnd((C,), torch.float32, 4)
    out = ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda())
    check('layernorm', out, F.layer_norm(x.float(), (C,), gamma, beta, 1e-5), dtype)


def attn_ref(q, k, v, B, Lq, Lk, heads, d):
    q = q.float().reshape(B, Lq, heads, d).transpose(1, 2)
    k = k.float().reshape(B, Lk, heads, d).transpose(1, 2)
    v = v.float().reshape(B, Lk, heads, d).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, -1)
    return (p @ v).transpose(1, 2).reshape(B * Lq, heads * d)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('B,Lq,Lk,heads,d', [(2, 256, 256, 8, 40), (1, 4096, 4096, 2, 40), (2, 1024, 1024, 8, 80),
                                             (2, 256, 256, 8, 160), (3, 64, 64, 8, 160), (2, 1000, 77, 8, 40),
                                             (2, 256, 77, 8, 80), (1, 64, 93, 8, 160), (2, 300, 300, 5, 64),
                                             (1, 128, 16, 8, 40)])
def test_attention(lib, dtype, B, Lq, Lk, heads, d):
    from mvedit_amd import ops
    C = heads * d
    q, k, v = rnd((B * Lq, C), dtype, 1), rnd((B * Lk, C), dtype, 2), rnd((B * Lk, C), dtype, 3)
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), B, Lq, Lk, heads, d)
    check('attention', out, attn_ref(q, k, v, B, Lq, Lk, heads, d), dtype, f'{(B, Lq, Lk, heads, d)}')


def test_attention_packed_qkv_spike_and_segments(lib):
    from mvedit_amd import ops
    dtype = torch.float16
    B, L, heads, d = 2, 320, 8, 40
    C = heads * d
    qkv = rnd((B * L, 3 * C), dtype, 1)
    # force the online-softmax rescale path: one key row dominates a late tile for one query
    qkv[5, :C] *= 6
    qkv[300, C:2 * C] = qkv[5, :C]
    g = qkv.cuda()
    out = ops.attention(g[:, :C], g[:, C:2 * C], g[:, 2 * C:], B, L, L, heads, d)
    check('attention packed qkv', out, attn_ref(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, L, L, heads, d), dtype)
    # two KV segments == concatenation along the key axis (reference attention)
    L2 = 150
    k2, v2 = rnd((B * L2, C), dtype, 4), rnd((B * L2, C), dtype, 5)
    kc = torch.cat([qkv[:, C:2 * C].reshape(B, L, C), k2.reshape(B, L2, C)], 1).reshape(-1, C)
    vc = torch.cat([qkv[:, 2 * C:].reshape(B, L, C), v2.reshape(B, L2, C)], 1).reshape(-1, C)
    out = ops.attention(g[:, :C], g[:, C:2 * C], g[:, 2 * C:], B, L, L, heads, d, k2=k2.cuda(), v2=v2.cuda(), Lk2=L2)
    check('attention 2 segments', out, attn_ref(qkv[:, :C], kc, vc, B, L, L + L2, heads, d), dtype)
    # cross-image attention (joint_attn.py:13-17): [2b, L, C] viewed as [b, 2L, C]
    out = ops.attention(g[:, :C], g[:, C:2 * C], g[:, 2 * C:], B // 2, 2 * L, 2 * L, heads, d)
    check('attention cross-image', out,
          attn_ref(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B // 2, 2 * L, 2 * L, heads, d), dtype)


@pytest.mark.parametrize('variant', [0, 8, 9, 10, 11, 12, 13, 14, 15])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_attention_d40_transposed_v_kernel(lib, dtype, variant):
    """k_attention3 (mve_attention_tune(8..15): 4 or 8 waves per block, 2 to 4 LDS stages, 12..15 with the software-pipelined tile loop;
    32x32x16 Q K^T, V staged row-major by LDS-DMA and read through ds_read_b64_tr_b16, P moved
    into its operand layout with v_permlane16_swap) on every d = 40 situation the UNet produces: full tiles, a ragged last key tile
    (Lk = 77 / 93 / 16 / 300), query counts that do not fill the 128-row block, the rescale path (a key dominating a late tile), strided
    packed-QKV views, two KV segments, cross-image pairing -- against fp32 torch with the per-kernel bar of test_attention."""
    from mvedit_amd import _lib, ops
    tune = _lib.raw('mve_attention_tune')
    old = tune(-1)
    try:
        tune(variant)
        heads, d = 8, 40
        C = heads * d
        for B, Lq, Lk in [(2, 256, 256), (1, 4096, 4096), (2, 1000, 77), (1, 128, 16), (2, 300, 300), (3, 576, 93), (1, 64, 4096)]:
            q, k, v = rnd((B * Lq, C), dtype, 1), rnd((B * Lk, C), dtype, 2), rnd((B * Lk, C), dtype, 3)
            out = ops.attention(q.cuda(), k.cuda(), v.cuda(), B, Lq, Lk, heads, d)
            check(f'attention variant {variant}', out, attn_ref(q, k, v, B, Lq, Lk, heads, d), dtype, f'{(B, Lq, Lk)}')
        B, L = 2, 320
        qkv = rnd((B * L, 3 * C), dtype, 1)
        qkv[5, :C] *= 6
        qkv[300, C:2 * C] = qkv[5, :C]
        g = qkv.cuda()
        out = ops.attention(g[:, :C], g[:, C:2 * C], g[:, 2 * C:], B, L, L, heads, d)
        check('variant packed qkv + spike', out, attn_ref(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, L, L, heads, d), dtype)
        L2 = 150
        k2, v2 = rnd((B * L2, C), dtype, 4), rnd((B * L2, C), dtype, 5)
        kc = torch.cat([qkv[:, C:2 * C].reshape(B, L, C), k2.reshape(B, L2, C)], 1).reshape(-1, C)
        vc = torch.cat([qkv[:, 2 * C:].reshape(B, L, C), v2.reshape(B, L2, C)], 1).reshape(-1, C)
        out = ops.attention(g[:, :C], g[:, C:2 * C], g[:, 2 * C:], B, L, L, heads, d, k2=k2.cuda(), v2=v2.cuda(), Lk2=L2)
        check('variant 2 segments', out, attn_ref(qkv[:, :C], kc, vc, B, L, L + L2, heads, d), dtype)
        out = ops.attention(g[:, :C], g[:, C:2 * C], g[:, 2 * C:], B // 2, 2 * L, 2 * L, heads, d)
        check('variant cross-image', out, attn_ref(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B // 2, 2 * L, 2 * L, heads, d), dtype)
        # Q carrying softmax_scale * log2(e) (how the executors call the kernel): the maximum is subtracted by the MFMA's C operand
        c = d ** -0.5 * math.log2(math.e)
        for Bp, Lqp, Lkp in [(2, 256, 256), (1, 2048, 2048), (2, 1000, 77), (1, 128, 16), (3, 576, 93)]:
            q, k, v = rnd((Bp * Lqp, C), dtype, 11), rnd((Bp * Lkp, C), dtype, 12), rnd((Bp * Lkp, C), dtype, 13)
            qp = (q.float() * c).to(dtype)
            out = ops.attention(qp.cuda(), k.cuda(), v.cuda(), Bp, Lqp, Lkp, heads, d, prescaled=True)
            check(f'attention variant {variant} prescaled', out, attn_ref(qp.float() / c, k, v, Bp, Lqp, Lkp, heads, d), dtype, f'{(Bp, Lqp, Lkp)}')
        qp = (qkv[:, :C].float() * c).to(dtype)          # the spike row: rescale path of the prescaled kernel; second KV segment
        out = ops.attention(qp.cuda(), g[:, C:2 * C], g[:, 2 * C:], B, L, L, heads, d, k2=k2.cuda(), v2=v2.cuda(), Lk2=L2, prescaled=True)
        check('variant prescaled spike + 2 segments', out, attn_ref(qp.float() / c, kc, vc, B, L, L + L2, heads, d), dtype)
        # all logits far below zero in the first tile (the prescaled kernel starts from m = 0): no underflow of the denominator
        q, k, v = rnd((128, C), dtype, 21), rnd((128, C), dtype, 22), rnd((128, C), dtype, 23)
        k = (k.float().abs() + 2).to(dtype)
        qp = (-(q.float().abs() + 2) * c).to(dtype)
        out = ops.attention(qp.cuda(), k.cuda(), v.cuda(), 1, 128, 128, heads, d, prescaled=True)
        check('variant prescaled negative logits', out, attn_ref(qp.float() / c, k, v, 1, 128, 128, heads, d), dtype)
        # a non-default softmax scale and a value spread that makes the ones-row denominator matter
        q, k, v = rnd((256, C), dtype, 7) * 3, rnd((256, C), dtype, 8) * 3, rnd((256, C), dtype, 9) + 2
        out = ops.attention(q.cuda(), k.cuda(), v.cuda(), 1, 256, 256, heads, d)
        check('variant large logits', out, attn_ref(q, k, v, 1, 256, 256, heads, d), dtype)
    finally:
        tune(old)


def test_boundary_helpers(lib):
    from mvedit_amd import ops
    x = rnd((3, 4, 8, 6), torch.float32, 1)
    y = ops.nchw_to_nhwc(x.cuda(), torch.float16)
    assert y.shape == (3 * 48, 8)
    ref = torch.zeros(3, 8, 6, 8)
    ref[..., :4] = x.permute(0, 2, 3, 1)
    assert torch.equal(y.cpu().float(), ref.reshape(-1, 8).half().float())
    back = ops.nhwc_to_nchw(y, 3, 4, 8, 6, torch.float32)
    assert torch.equal(back.cpu(), x.half().float())
    t = torch.tensor([499.0, 3.0, 981.0])
    emb = ops.timestep_embedding(t.cuda(), 320, torch.float16).float().cpu()
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half) / half)
    arg = t[:, None] * freqs[None]
    ref = torch.cat([torch.cos(arg), torch.sin(arg)], -1)
    assert (emb - ref).abs().max() < 2e-3
    a, b = rnd((64, 320), torch.float16, 2), rnd((64, 320), torch.float16, 3)
    assert torch.allclose(ops.axpy(a.cuda(), b.cuda(), 0.5).float().cpu(), (a.float() + 0.5 * b.float()).half().float())
    assert torch.allclose(ops.silu(a.cuda()).float().cpu(), F.silu(a.float()).half().float(), atol=2e-3)
    u, tx = rnd((2, 4, 8, 8), torch.float32, 4), rnd((2, 4, 8, 8), torch.float32, 5)
    assert torch.allclose(ops.cfg_combine(u.cuda(), tx.cuda(), 7.0).cpu(), 7.0 * tx + (1 - 7.0) * u, atol=1e-5)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_splitk_small_m_big_k(lib, dtype):
    """Deep UNet levels at small batch: M = B*64 rows, K = 9*1280.  The K range is cut into concurrent slices whose fp32
    partials are summed in a fixed order; results must match the unsplit kernel to rounding and be deterministic."""
    from mvedit_amd import ops, _lib
    M, N, K = 512, 1280, 11520
    ws = _lib.raw('mve_gemm_workspace_bytes')
    assert ws(M, N, K, 64) > 0 and ws(262144, 320, 2880, 4096) == 0 and ws(M, N, K, 0) == 0
    assert ws(8 * M, N, K, 64) == 8 * ws(M, N, K, 64)          # the slice count does not depend on the batch
    a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    bias, res = rnd((N,), torch.float32, 3), rnd((M, N), dtype, 4)
    ref = a.float() @ w.float().t() + bias + res.float()
    o1 = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), rows_per_image=64)
    o0 = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda())
    check('gemm split-K', o1, ref, dtype)
    check('gemm no split', o0, ref, dtype)
    assert torch.equal(o1, ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), rows_per_image=64))
    # batch invariance: the first image's rows are bit-identical when it is processed alone
    assert torch.equal(o1[:64], ops.gemm(a[:64].cuda(), w.cuda(), bias=bias.cuda(), residual=res[:64].cuda(), rows_per_image=64))
    # conv at 8x8, batch 2 (M = 128): tap-major and slab-major weight layouts, K slices start mid-way through the taps
    B, H, C = 2, 8, 1280
    x = rnd((B, C, H, H), dtype, 5)
    wc = rnd((C, C, 3, 3), dtype, 6, (9 * C) ** -0.5)
    refc = to_nhwc(conv_ref(x, wc, bias))
    for chunk64 in (False, True):
        w_k, wflag = ops.pack_conv_weight(wc, chunk64)
        out, _, _ = ops.conv3x3(to_nhwc(x).cuda(), w_k.cuda(), B, H, H, bias=bias.cuda(), flags=wflag, splitk=True)
        check('conv split-K', out, refc, dtype, f'chunk64={chunk64}')
    # GEGLU through the reducer
    wv, wg = rnd((640, K), dtype, 7, K ** -0.5), rnd((640, K), dtype, 8, K ** -0.5)
    w_il = torch.stack([wv, wg], 1).reshape(1280, K).contiguous()
    out = ops.gemm(a.cuda(), w_il.cuda(), flags=ops.GEGLU, rows_per_image=64)
    check('geglu split-K', out, (a.float() @ wv.float().t()) * F.gelu(a.float() @ wg.float().t()), dtype)


@pytest.mark.gpu
def test_big_tile_kernel_is_bit_identical(lib):
    """The 256x320 kernel (csrc/gemm_big.hip) walks K in the same order with the same MFMA as the 128x160 kernel, so the
    dispatcher's size-based choice must never change a single bit (it would break batch / partition invariance)."""
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        dtype = torch.float16
        for (M, N, K, rpi, fl) in [(4096, 640, 640, 0, 0), (1000, 320, 328, 0, 0), (2048, 2560, 320, 0, ops.GEGLU), (512, 1280, 11520, 64, 0)]:
            a, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
            bias = rnd((N,), torch.float32, 3).cuda()
            res = None if fl else rnd((M, N), dtype, 4).cuda()
            outs = []
            for big in TILE_MODES:
                tune(big)
                outs.append(ops.gemm(a, w, bias=bias, residual=res, flags=fl, rows_per_image=rpi))
            assert all(torch.equal(outs[0], o) for o in outs[1:]), (M, N, K)
        for (B, H, C1, C2, Cout, stride, ups) in [(4, 16, 320, 0, 320, 1, False), (3, 16, 640, 320, 640, 1, False), (2, 16, 320, 0, 320, 2, False),
                                                  (2, 8, 640, 0, 640, 1, True), (5, 9, 72, 0, 320, 1, False)]:
            x1 = to_nhwc(rnd((B, C1, H, H), dtype, 1)).cuda()
            x2 = to_nhwc(rnd((B, C2, H, H), dtype, 2)).cuda() if C2 else None
            wt = rnd((Cout, C1 + C2, 3, 3), dtype, 3, (9 * (C1 + C2)) ** -0.5)
            w_k, wflag = ops.pack_conv_weight(wt, (C1 % 64 == 0 and C2 % 64 == 0))
            outs = []
            for big in TILE_MODES:
                tune(big)
                outs.append(ops.conv3x3(x1, w_k.cuda(), B, H, H, x2=x2, stride=stride, upsample=ups, flags=wflag)[0])
            assert all(torch.equal(outs[0], o) for o in outs[1:]), (B, H, C1, C2, Cout, stride, ups)
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,C1,C3,C4', [(2, 16, 640, 320, 0), (3, 8, 320, 640, 320), (2, 8, 1280, 1280, 1280)])
def test_conv3x3_with_fused_shortcut(lib, B, H, C1, C3, C4):
    """conv2 + 1x1 conv_shortcut over the (concatenated) block input in one K loop, both kernels (bit-identical), with split-K."""
    from mvedit_amd import ops, _lib
    dtype = torch.float16
    Cout = C1
    h = rnd((B, C1, H, H), dtype, 1)
    x3 = rnd((B, C3, H, H), dtype, 2)
    x4 = rnd((B, C4, H, H), dtype, 3) if C4 else None
    w = rnd((Cout, C1, 3, 3), dtype, 4, (9 * C1) ** -0.5)
    wsc = rnd((Cout, C3 + C4, 1, 1), dtype, 5, (C3 + C4) ** -0.5)
    b2, bs = rnd((Cout,), torch.float32, 6), rnd((Cout,), torch.float32, 7)
    xin = torch.cat([x3, x4], 1) if C4 else x3
    ref = conv_ref(h, w, b2) + F.conv2d(xin.float(), wsc.float(), bs)
    w_k, _ = ops.pack_conv_weight(w, True)
    wcat = torch.cat([w_k.reshape(Cout, -1), wsc.reshape(Cout, -1)], dim=1).contiguous()
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        outs = []
        for big in TILE_MODES:
            tune(big)
            outs.append(ops.conv3x3_shortcut(to_nhwc(h).cuda(), wcat.cuda(), B, H, H, to_nhwc(x3).cuda(),
                                             to_nhwc(x4).cuda() if C4 else None, bias=b2.cuda(), bias2=bs.cuda()))
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        check('conv3x3 + shortcut', outs[0], to_nhwc(ref), dtype)
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,C,Cout', [(2, 16, 12, 64, 64), (1, 64, 64, 128, 320), (2, 8, 8, 32, 32)])
def test_conv3x3_downsample_pad_bottom_right(lib, B, H, W, C, Cout):
    """diffusers Downsample2D(padding=0) of the VAE encoder: F.pad(x, (0, 1, 0, 1)) then a stride-2 conv without padding
    (MVE_CONV_PAD_BR); both weight layouts, both kernels bit-identical."""
    from mvedit_amd import ops, _lib
    dtype = torch.float16
    x = rnd((B, C, H, W), dtype, 1)
    w = rnd((Cout, C, 3, 3), dtype, 2, (9 * C) ** -0.5)
    bias = rnd((Cout,), torch.float32, 3)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), bias, stride=2)
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        for chunk64 in ([False, True] if C % 64 == 0 else [False]):
            w_k, wflag = ops.pack_conv_weight(w, chunk64)
            outs = []
            for big in TILE_MODES:
                tune(big)
                out, Ho, Wo = ops.conv3x3(to_nhwc(x).cuda(), w_k.cuda(), B, H, W, stride=2, bias=bias.cuda(), flags=wflag | ops.PAD_BR)
                assert (Ho, Wo) == (H // 2, W // 2)
                outs.append(out)
            assert all(torch.equal(outs[0], o) for o in outs[1:])
            check('conv3x3 pad_br', outs[0], to_nhwc(ref), dtype, f'chunk64={chunk64}')
    finally:
        tune(old)
    with pytest.raises(_lib.MveError):      # the flag only exists for the stride-2 downsampler
        ops.conv3x3(to_nhwc(x).cuda(), w_k.cuda(), B, H, W, stride=1, bias=bias.cuda(), flags=wflag | ops.PAD_BR)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('M,N', [(64, 128), (7, 4096), (3, 1028)])
def test_softmax_rows(lib, dtype, M, N):
    from mvedit_amd import ops
    s = rnd((M, N), torch.float32, 1) * 6
    s[0, :4] = 60.0                                               # a dominated row: everything else underflows to zero
    out = ops.softmax_rows(s.cuda(), dtype)
    ref = torch.softmax(s, dim=-1)
    assert out.dtype == dtype and out.shape == (M, N)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert ((out.float().cpu() - ref).abs() <= eps * ref + 1e-7).all()
    assert (out.float().sum(-1).cpu() - 1).abs().max() < 4 * eps


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('B,H,W,C1,C2,Cout,stride,ups', [
    (2, 10, 6, 128, 0, 320, 1, True),        # Upsample2D through the per-row fast addressing, odd-ish non-square size
    (1, 16, 16, 64, 64, 256, 1, True),       # ... with a concatenated input, 256-wide big tile
    (2, 16, 16, 128, 0, 256, 1, False),      # 256-wide big tile (the VAE's widths), plain
    (2, 16, 16, 128, 0, 512, 2, False),
])
def test_conv3x3_big_tiles_and_upsample_addressing(lib, dtype, B, H, W, C1, C2, Cout, stride, ups):
    """The 256 x {320, 256} kernels against the 128-row kernel (bit-identical, forced through mve_gemm_tune) and the reference:
    nearest-2x upsample in the slab-major fast addressing path, and the 256-wide tile used for N = 256 / 512."""
    from mvedit_amd import ops, _lib
    x1 = rnd((B, C1, H, W), dtype, 1)
    x2 = rnd((B, C2, H, W), dtype, 2) if C2 else None
    C = C1 + C2
    w = rnd((Cout, C, 3, 3), dtype, 3, (9 * C) ** -0.5)
    bias = rnd((Cout,), torch.float32, 4)
    ref = conv_ref(torch.cat([x1, x2], 1) if C2 else x1, w, bias, stride, ups)
    w_k, wflag = ops.pack_conv_weight(w, True)
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        outs = []
        for big in TILE_MODES:
            tune(big)
            out, Ho, Wo = ops.conv3x3(to_nhwc(x1).cuda(), w_k.cuda(), B, H, W, x2=to_nhwc(x2).cuda() if C2 else None, stride=stride,
                                      upsample=ups, bias=bias.cuda(), flags=wflag)
            outs.append(out)
        assert (Ho, Wo) == tuple(ref.shape[2:])
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        check('conv3x3 big/upsample', outs[0], to_nhwc(ref), dtype, f'{(B, H, W, C1, C2, Cout, stride, ups)}')
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('B,H,W,C,Cout', [
    (2, 8, 8, 128, 320),          # one partial 256-row tile per phase
    (4, 16, 16, 64, 128),         # the 128-wide tile (VAE, image resolution), whole tiles, one 64-channel slab
    (1, 32, 32, 64, 256),         # the 256-wide tile; rows of a tile span 8 source rows
    (3, 16, 8, 192, 640),         # non-square, W = 8: a tile spans 32 source rows, ragged last tile
    (2, 8, 8, 1280, 1280),        # the 8 x 8 level of the UNet: K = 5120 in slices + reducer
    (1, 4, 256, 64, 128),         # a source row longer than a tile would be split; here W = 256 = one tile per source row
    (8, 8, 8, 1280, 1280),        # 8 images at the 8 x 8 level: the four phases in ONE launch of 2048 rows, K in slices + one reducer
])
def test_upsample_conv_phases(lib, dtype, B, H, W, C, Cout):
    """Upsample2D as four 2 x 2 phase convs (mve_upsample_conv_phases): (a) the packed summed taps are the oracle's, bit for bit; (b) the output against
    the oracle's phase form on those rounded weights: the usual conv tolerance; (c) against nearest-2x + conv 3x3 on the original weights (what the
    reference computes): the extra weight rounding stays inside the same tolerance; (d) the residual-pair output."""
    from mvedit_amd import ops
    from oracle import unet_oracle as UO
    x = rnd((B, C, H, W), dtype, 1)
    w = rnd((Cout, C, 3, 3), dtype, 3, (9 * C) ** -0.5)
    bias = rnd((Cout,), torch.float32, 4)
    assert ops.upsample_conv_phases_supported(C, Cout, B, H, W)
    w4 = ops.pack_upsample_phase_weights(w.cuda())
    q = UO.quantizer(dtype)
    ref4 = UO.upsample_phase_weights(w, q)
    for py in (0, 1):
        for px in (0, 1):
            want = ref4[py][px].reshape(Cout, C // 64, 64, 2, 2).permute(0, 1, 3, 4, 2).to(dtype)        # [O][I/64][2][2][64]
            assert torch.equal(w4[2 * py + px].cpu(), want), (py, px)
    out = ops.upsample_conv_phases(to_nhwc(x).cuda(), w4, B, H, W, bias=bias.cuda())
    assert out.shape == (B * 4 * H * W, Cout)
    if (B * H * W) % 256 == 0:        # that was one launch for the four phases: four launches give the same bits
        from mvedit_amd import _lib
        old = _lib.raw('mve_upsample_conv_phases_tune')(0)
        try:
            out4 = ops.upsample_conv_phases(to_nhwc(x).cuda(), w4, B, H, W, bias=bias.cuda())
        finally:
            _lib.raw('mve_upsample_conv_phases_tune')(old)
        assert torch.equal(out, out4)          # (also where K is sliced: the slice rule looks at the image, not at the launch)
    check('upsample phases vs oracle phases', out, to_nhwc(UO.upsample_conv_phases(x, w, bias, q)), dtype, f'{(B, H, W, C, Cout)}')
    check('upsample phases vs upsample + conv3x3', out, to_nhwc(conv_ref(x, w, bias, 1, True)), dtype, f'{(B, H, W, C, Cout)}')
    if Cout % 320 == 0:
        hi, lo = ops.upsample_conv_phases(to_nhwc(x).cuda(), w4, B, H, W, bias=bias.cuda(), pair_out=True)
        v = to_nhwc(UO.upsample_conv_phases(x, w, bias, q))
        check('upsample phases, pair hi', hi, v, dtype)
        err_pair = float(((hi.double().cpu() + _lo(lo)) - v.double()).norm() / v.double().norm())
        err_hi = float((hi.float().cpu() - v).norm() / v.norm())
        assert err_pair < 0.25 * err_hi, (err_pair, err_hi)


@pytest.mark.gpu
def test_gemm_256_wide_tile_matches_small_kernel(lib):
    from mvedit_amd import ops, _lib
    dtype = torch.float16
    M, N, K = 700, 512, 328
    a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    bias = rnd((N,), torch.float32, 3)
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        outs = []
        for big in TILE_MODES:
            tune(big)
            outs.append(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda()))
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        check('gemm 256-wide', outs[0], a.float() @ w.float().t() + bias, dtype)
    finally:
        tune(old)


@pytest.mark.gpu
def test_attention_experimental_variant_matches_default(lib):
    """mve_attention_tune(1): 16 query rows per wave / 64-key fills for head dim 40 (an A/B candidate, tools/ab_attention.py).  Same
    function, different rescale points: agreement to rounding with the default and with the reference softmax."""
    from mvedit_amd import ops, _lib
    dtype = torch.float16
    B, L, heads, d = 2, 333, 8, 40
    qkv = rnd((B * L, 3 * heads * d), dtype, 1)
    q, k, v = (qkv[:, i * heads * d:(i + 1) * heads * d].cuda() for i in range(3))
    tune = _lib.raw('mve_attention_tune')
    old = tune(-1)
    try:
        tune(0)
        o0 = ops.attention(q, k, v, B, L, L, heads, d)
        tune(1)
        o1 = ops.attention(q, k, v, B, L, L, heads, d)
    finally:
        tune(old)
    qf, kf, vf = (t.float().cpu().reshape(B, L, heads, d).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(B * L, heads * d)
    check('attention variant 1', o1, ref, dtype)
    assert (o1.float() - o0.float()).abs().max() < 2e-3


@pytest.mark.gpu
def test_attention_conflict_free_k_swizzle_is_bit_identical(lib):
    """mve_attention_tune(2): the K tile of head dims 80 / 160 stored under a chunk permutation whose fragment reads have no LDS bank
    conflicts (derived analytically from the ds_read_b128 lane groups; tools/lds_conflicts.py).  Pure data-layout change: bitwise equal."""
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_attention_tune')
    old = tune(-1)
    try:
        for heads, d, L in ((8, 80, 300), (8, 160, 150)):
            qkv = rnd((2 * L, 3 * heads * d), torch.float16, d)
            q, k, v = (qkv[:, i * heads * d:(i + 1) * heads * d].cuda() for i in range(3))
            tune(0)
            o0 = ops.attention(q, k, v, 2, L, L, heads, d)
            tune(2)
            o2 = ops.attention(q, k, v, 2, L, L, heads, d)
            assert torch.equal(o0, o2), (heads, d)
    finally:
        tune(old)


@pytest.mark.gpu
def test_attention_vt_store_swizzle_is_bit_identical(lib):
    """mve_attention_tune(4) / (6): the V^T tile under a swizzle that also spreads the transposing ds_write_b32 stores over the banks
    (tools/lds_conflicts.py: 5- to 10-way -> 2- to 4-way, reads still conflict-free).  Pure data-layout change: bitwise equal."""
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_attention_tune')
    old = tune(-1)
    try:
        for dtype in (torch.float16, torch.bfloat16):
            for heads, d, L in ((8, 40, 333), (5, 64, 257), (8, 80, 300), (8, 160, 150)):
                qkv = rnd((2 * L, 3 * heads * d), dtype, d)
                q, k, v = (qkv[:, i * heads * d:(i + 1) * heads * d].cuda() for i in range(3))
                tune(0)
                o0 = ops.attention(q, k, v, 2, L, L, heads, d)
                for variant in ((4,) if d in (40, 64) else (4, 6)):
                    tune(variant)
                    o1 = ops.attention(q, k, v, 2, L, L, heads, d)
                    assert torch.equal(o0, o1), (dtype, heads, d, variant)
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_pingpong_main_loop_is_bit_identical_and_race_free(lib, dtype):
    """csrc/gemm_pp.hip keeps LDS-DMA in flight across barriers with counted vmcnt and a 4-slot ring: a slot read too early or
    overwritten too early shows up as a wrong tile that comes and goes.  Many-block launches with long and short K, repeated, against
    the two-stage kernel (same MFMA order -> identical bits): dense, GEGLU, split-K (parallel and sequential), conv with concat /
    shortcut / halo rows, M and N tails."""
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
      for strict in (0, STRICT):          # one accumulation chain (default) / emulated K slices: the two loops agree bitwise under either rule
        PP, BIG = 1 | strict, 1 | (1 << 27) | strict
        for (M, N, K, rpi, fl) in [(70000, 320, 320, 0, 0), (33000, 960, 64, 0, 0), (20000, 2560, 320, 0, ops.GEGLU), (16384, 320, 1280, 0, 0),
                                   (9000, 1280, 5120, 0, 0), (1024, 1280, 11520, 64, 0), (777, 512, 640, 0, 0), (4096, 256, 4608, 0, 0),
                                   (20000, 128, 1152, 0, 0), (5000, 384, 640, 0, 0)]:      # 128-wide tile (4 x 2 waves)
            a, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
            bias = rnd((N,), torch.float32, 3).cuda()
            res = None if fl else rnd((M, N), dtype, 4).cuda()
            tune(BIG)
            ref = ops.gemm(a, w, bias=bias, residual=res, flags=fl, rows_per_image=rpi)
            tune(PP)
            for rep in range(4):
                out = ops.gemm(a, w, bias=bias, residual=res, flags=fl, rows_per_image=rpi)
                assert torch.equal(out, ref), (M, N, K, rep, int((out != ref).sum()))
        for (B, H, C1, C2, Cout, stride, ups) in [(24, 32, 320, 0, 320, 1, False), (8, 32, 640, 320, 640, 1, False), (6, 32, 320, 0, 640, 2, False),
                                                  (4, 16, 640, 0, 640, 1, True), (16, 8, 1280, 1280, 1280, 1, False), (2, 64, 128, 0, 256, 1, False),
                                                  (3, 24, 192, 64, 512, 1, True),
                                                  (2, 64, 128, 0, 128, 1, False), (2, 32, 256, 0, 128, 1, True), (3, 48, 64, 64, 128, 2, False)]:
            x1 = to_nhwc(rnd((B, C1, H, H), dtype, 1)).cuda()
            x2 = to_nhwc(rnd((B, C2, H, H), dtype, 2)).cuda() if C2 else None
            wt = rnd((Cout, C1 + C2, 3, 3), dtype, 3, (9 * (C1 + C2)) ** -0.5)
            w_k, wflag = ops.pack_conv_weight(wt, True)
            w_k = w_k.cuda()
            for sk in (False, True):
                tune(BIG)
                ref = ops.conv3x3(x1, w_k, B, H, H, x2=x2, stride=stride, upsample=ups, flags=wflag, splitk=sk)[0]
                tune(PP)
                for rep in range(3):
                    out = ops.conv3x3(x1, w_k, B, H, H, x2=x2, stride=stride, upsample=ups, flags=wflag, splitk=sk)[0]
                    assert torch.equal(out, ref), (B, H, C1, C2, Cout, stride, ups, sk, rep, int((out != ref).sum()))
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_unsplit_chain_vs_sliced_sum(lib, dtype):
    """Round 4: a launch that fills the chip un-split accumulates K in ONE chain by default instead of emulating the K slices of the slice rule
    (csrc/gemm.hip gemm_strict_splitk).  The two differ by fp32 summation order only: every element within one rounding step of the
    16-bit output of the other, a small fraction of elements differing at all, and both inside the kernel bar against the fp32 reference.
    (mve_gemm_tune threshold 1 makes a test-sized launch 'fill the chip'; bit 30 selects the emulated slices.)"""
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    try:
        for (M, N, K, rpi) in [(1024, 1280, 11520, 64), (2048, 640, 2560, 1024), (1024, 1280, 5120, 256)]:
            a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
            bias, res = rnd((N,), torch.float32, 3), rnd((M, N), dtype, 4)
            assert _lib.raw('mve_gemm_workspace_bytes')(M, N, K, rpi) > 0, 'the slice rule must ask for slices here'
            tune(1)
            chain = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), rows_per_image=rpi)
            tune(1 | STRICT)
            sliced = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), rows_per_image=rpi)
            tune(0)
            small = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), rows_per_image=rpi)      # 128-row kernel: real split-K + reducer
            assert torch.equal(sliced, small), 'the emulated slices equal split-K + reducer bitwise'
            d = (chain.float() - sliced.float()).abs()
            # (+ 8e-6: where bias + residual cancel the sum, the fp32 difference of the two chains -- ~2^-22 of the O(1) terms -- spans several 16-bit steps)
            bound = ulp * torch.maximum(chain.float().abs(), sliced.float().abs()) + 8e-6
            frac = float((d > 0).float().mean())
            print(f'{dtype} M={M} N={N} K={K}: {frac * 100:.2f} % of the elements differ between one chain and the sliced sum, max {float((d / bound).max()):.2f} output roundings')
            assert bool((d <= bound).all()) and frac < 0.2, (M, N, K, frac)
            ref = a.float() @ w.float().t() + bias + res.float()
            check('gemm, one chain', chain, ref, dtype)
            check('gemm, sliced', sliced, ref, dtype)
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_pingpong_160_wide_tile_for_small_batches(lib, dtype):
    """When 256 x 320 tiles would leave CUs without a block the dispatcher halves the tile width (256 x 160, waves 4 x 2, the 10 weight
    pieces of a step split 3 / 3 / 2 / 2 with a zero-fill filler): same K order, same bits as the 128-row kernel.  The block threshold
    is set so that the 320-wide tiling falls short and the 160-wide one does not."""
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        for (M, N, K, fl, thr) in [(2048, 320, 640, 0, 12), (2000, 640, 320, 0, 20), (1024, 2560, 320, ops.GEGLU, 40), (4096, 320, 2560, 0, 24)]:
            a, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
            bias = rnd((N,), torch.float32, 3).cuda()
            res = None if fl else rnd((M, N), dtype, 4).cuda()
            tune(0)
            ref = ops.gemm(a, w, bias=bias, residual=res, flags=fl)
            tune(thr)
            for rep in range(3):
                out = ops.gemm(a, w, bias=bias, residual=res, flags=fl)
                assert torch.equal(out, ref), (M, N, K, rep, int((out != ref).sum()))
        for (B, H, C1, C2, Cout, stride, ups, thr) in [(2, 32, 320, 0, 320, 1, False, 12), (1, 32, 640, 320, 640, 1, False, 12), (2, 16, 320, 0, 320, 1, True, 12),
                                                       (3, 32, 64, 0, 320, 2, False, 5)]:
            x1 = to_nhwc(rnd((B, C1, H, H), dtype, 1)).cuda()
            x2 = to_nhwc(rnd((B, C2, H, H), dtype, 2)).cuda() if C2 else None
            wt = rnd((Cout, C1 + C2, 3, 3), dtype, 3, (9 * (C1 + C2)) ** -0.5)
            w_k, wflag = ops.pack_conv_weight(wt, True)
            w_k = w_k.cuda()
            bias = rnd((Cout,), torch.float32, 4).cuda()
            tune(0)
            ref = ops.conv3x3(x1, w_k, B, H, H, x2=x2, stride=stride, upsample=ups, bias=bias, flags=wflag, splitk=False)[0]
            tune(thr)
            for rep in range(3):
                out = ops.conv3x3(x1, w_k, B, H, H, x2=x2, stride=stride, upsample=ups, bias=bias, flags=wflag, splitk=False)[0]
                assert torch.equal(out, ref), (B, H, C1, C2, Cout, stride, ups, rep, int((out != ref).sum()))
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('M,N,K,geglu,res', [(8 * 4096, 320, 320, False, True), (8 * 4096, 960, 320, False, False), (8 * 4096, 2560, 320, True, False),
                                              (8 * 4096, 320, 1280, False, True), (16 * 1024, 640, 640, False, True)])
def test_gemm_narrow_launches_on_the_two_block_tile_are_bit_identical(lib, dtype, M, N, K, geglu, res):
    """Launches too small to fill the chip with 320-wide tiles take the two-blocks-per-CU 256 x 160 three-slot tile by default
    (mve_gemm_tune: bit 28 turns it off, bit 26 forces it everywhere): same bits as the four-slot tile and as the 128-row kernel."""
    from mvedit_amd import ops, _lib
    a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    bias = rnd((N,), torch.float32, 3)
    r = rnd((M, N // 2 if geglu else N), dtype, 4) if res else None
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        outs = []
        for word in (256, 256 | (1 << 28), 256 | (1 << 26), 0):          # default (narrow launches only) / never / everywhere / 128-row kernel
            tune(word)
            outs.append(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=r.cuda() if res else None, flags=ops.GEGLU if geglu else 0))
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        tune(256 | (1 << 28))
        assert tune(-1) == 256 | (1 << 28)                       # the whole word comes back: old = tune(x); ...; tune(old) restores every switch
    finally:
        tune(old)


def _split_pair(x32, dtype):
    """(hi, lo8) in the engine's stream-pair format: hi = round16(x), lo8 = E5M2(2^8 (x - hi)) as uint8 (include/mvedit_amd.h)."""
    from mvedit_amd import ops
    return ops.split_pair(x32, dtype)


def _lo(lo8):
    """the value of a low half"""
    from mvedit_amd import ops
    return ops.lo8_to_float(lo8.cpu()).double()


# what the 8-bit low half leaves of the rounding remainder: |x - hi| <= 2^-11 |x| (fp16) / 2^-8 |x| (bf16), of which E5M2 keeps two mantissa bits
# (relative error <= 2^-3): |x - (hi + lo8)| <= 2^-14 |x| / 2^-11 |x|
def _pair_tol(dtype):
    return 7e-5 if dtype == torch.float16 else 5.2e-4


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_gemm_residual_pair(lib, dtype):
    """mve_gemm_pair (the executor's residual_pair mode): the residual arrives as an unrounded (hi, lo) pair, the result leaves as one.
    hi + lo8 must reproduce the fp32 value a w^T + bias + residual to ~2^-14 of its magnitude (16-bit storage alone: 2^-11 / 2^-8; the low half
    is 8 bits since round 5), on the 256-row tile, the 128-row kernel (both start their accumulators from the residual) and through the split-K
    reducer; hi alone must be the correctly rounded value up to one rounding step; without companions the call is the plain one, bitwise."""
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    try:
        for (M, N, K, rpi) in [(1024, 320, 320, 0), (2048, 640, 2560, 0), (700, 320, 1280, 0), (512, 1280, 5120, 64)]:
            a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
            bias = rnd((N,), torch.float32, 3)
            r32 = torch.randn(M, N, generator=torch.Generator().manual_seed(9)) * 3
            rh, rl = _split_pair(r32, dtype)
            ref = a.double() @ w.double().t() + bias.double() + (rh.double() + _lo(rl))
            for word in (1, 0):                      # 256-row tile wherever it fits / 128-row kernel only
                tune(word)
                hi, lo = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=rh.cuda(), residual_lo=rl.cuda(), rows_per_image=rpi, pair_out=True)
                got = hi.double().cpu() + _lo(lo)
                err = float((got - ref).abs().max() / ref.abs().max())
                e_hi = float((hi.double().cpu() - ref).abs().max() / ref.abs().max())
                print(f'{dtype} M={M} N={N} K={K} tune={word}: |hi + lo - ref| / max|ref| = {err:.2e}   (hi alone {e_hi:.2e})')
                assert err < _pair_tol(dtype), (M, N, K, word, err)
                half_ulp = (2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8) * hi.float().abs().cpu() * 1.01 + 1e-7
                assert bool((_lo(lo).abs().float() <= half_ulp).all()), 'lo is the rounding remainder of hi: at most half a step of it'
                plain = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=rh.cuda(), rows_per_image=rpi)
                same = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=rh.cuda(), rows_per_image=rpi, pair_out=False, residual_lo=None)
                assert torch.equal(plain, same)
            # out_lo without a residual: the pair of the fp32 accumulator value
            tune(1)
            hi, lo = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), rows_per_image=rpi, pair_out=True)
            ref0 = a.double() @ w.double().t() + bias.double()
            assert float(((hi.double().cpu() + _lo(lo)) - ref0).abs().max() / ref0.abs().max()) < _pair_tol(dtype)
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_pair_launches_round_identically_on_every_tile(lib, dtype):
    """Round 5: the residual pair is the executor's DEFAULT mode, so batch / partition invariance must hold in it: a pair launch gives the same
    (hi, lo) BITS on the 256-row tile (k_gemm_pp<PAIR>: accumulators started from the residual) and on the 128-row kernel (round 5: started from
    the residual too) -- dense GEMM, 3 x 3 conv with a residual pair, conv + fused shortcut, with a per-image row vector -- and the conv results
    stay inside the 1e-3 bar of the fp32 reference.  Without a residual the high half is the plain launch's output bit for bit."""
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_gemm_tune')
    old = tune(-1)
    g = torch.Generator().manual_seed(17)
    try:
        def both(fn):
            outs = []
            for word in (1, 0):                      # 256-row tile wherever it fits / 128-row kernel only
                tune(word)
                outs.append(fn())
            return outs
        # dense GEMM with a residual pair, bias
        for (M, N, K) in [(4096, 320, 320), (2048, 640, 2560), (1024, 1280, 1280), (65536, 320, 320)]:      # (the last: one tile per CU -- what
            # round 4's pair epilogue needed to leave stale LDS contents in ~1e-5 of the elements, profiles/r05_debug_pair_ops_v2.log)
            a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
            bias = rnd((N,), torch.float32, 3)
            rh, rl = _split_pair(torch.randn(M, N, generator=g) * 3, dtype)
            (h1, l1), (h0, l0) = both(lambda: ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=rh.cuda(), residual_lo=rl.cuda(), pair_out=True))
            assert torch.equal(h1, h0) and torch.equal(l1, l0), ('gemm', M, N, K, int((h1 != h0).sum()), int((l1 != l0).sum()))
        # 3 x 3 conv (slab-major weights) with a residual pair, bias and a per-image row vector
        for (B, H, C, Cout) in [(4, 32, 320, 320), (2, 32, 640, 640), (16, 8, 1280, 1280), (16, 64, 320, 320)]:
            x = rnd((B * H * H, C), dtype, 4)
            w, fl = ops.pack_conv_weight(rnd((Cout, C, 3, 3), dtype, 5, (9 * C) ** -0.5))
            bias, rv = rnd((Cout,), torch.float32, 6), rnd((B, Cout), torch.float32, 7)
            r32 = torch.randn(B * H * H, Cout, generator=g) * 2
            rh, rl = _split_pair(r32, dtype)
            run = lambda: ops.conv3x3(x.cuda(), w.cuda(), B, H, H, bias=bias.cuda(), rowvec=rv.cuda(), residual=rh.cuda(), residual_lo=rl.cuda(),
                                      flags=fl, splitk=False, pair_out=True)[0]
            (h1, l1), (h0, l0) = both(run)
            assert torch.equal(h1, h0) and torch.equal(l1, l0), ('conv', B, H, C, Cout, int((h1 != h0).sum()), int((l1 != l0).sum()))
            xi = x.float().view(B, H, H, C).permute(0, 3, 1, 2)
            wi = w.float().view(Cout, C // 64, 3, 3, 64).permute(0, 1, 4, 2, 3).reshape(Cout, C, 3, 3)
            ref = F.conv2d(xi, wi, bias, padding=1) + rv[:, :, None, None]
            ref = ref.double().permute(0, 2, 3, 1).reshape(B * H * H, Cout) + (rh.double() + _lo(rl))
            got = h1.double().cpu() + _lo(l1)
            assert float((got - ref).norm() / ref.norm()) < 1e-3, ('conv', B, H, C, Cout)
            assert float((got - ref).abs().max() / ref.abs().max()) < 1.5 * _pair_tol(dtype)
        # conv + fused 1 x 1 shortcut, pair output, no residual: hi == the plain launch, lo is its rounding remainder
        B, H, C1, C3, Cout = 4, 32, 640, 320, 640
        h2, x3 = rnd((B * H * H, Cout), dtype, 8), rnd((B * H * H, C3), dtype, 9)
        w3 = ops.pack_conv_weight(rnd((Cout, Cout, 3, 3), dtype, 10, (9 * Cout) ** -0.5))[0].reshape(Cout, -1)
        wsc = rnd((Cout, C3), dtype, 11, C3 ** -0.5)
        wcat = torch.cat([w3, wsc], 1).contiguous()
        b2, bs = rnd((Cout,), torch.float32, 12), rnd((Cout,), torch.float32, 13)
        (h1, l1), (h0, l0) = both(lambda: ops.conv3x3_shortcut(h2.cuda(), wcat.cuda(), B, H, H, x3.cuda(), bias=b2.cuda(), bias2=bs.cuda(), splitk=False, pair_out=True))
        assert torch.equal(h1, h0) and torch.equal(l1, l0)
        tune(1)
        plain = ops.conv3x3_shortcut(h2.cuda(), wcat.cuda(), B, H, H, x3.cuda(), bias=b2.cuda(), bias2=bs.cuda(), splitk=False)
        assert torch.equal(plain, h1)
    finally:
        tune(old)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_norms_read_the_residual_pair(lib, dtype):
    """mve_groupnorm_silu_pair / mve_layernorm_pair: statistics and normalisation over hi + lo equal the plain kernels run on an input whose
    16-bit rounding is exact (lo = 0: bitwise), and track the fp32 reference of the unrounded input more closely than the rounded input does."""
    from mvedit_amd import ops, _lib
    from mvedit_amd.ops import dt as _dt
    g = torch.Generator().manual_seed(5)
    B, HW, C, G = 2, 4096, 320, 32
    x32 = torch.randn(B * HW, C, generator=g) * 2 + 0.3
    hi, lo = _split_pair(x32, dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    ref = F.group_norm(x32.view(B, HW, C).permute(0, 2, 1).double(), G, gamma.double(), beta.double(), 1e-5)
    ref = F.silu(ref).permute(0, 2, 1).reshape(B * HW, C)
    dev = torch.device('cuda:0')
    gamma_d, beta_d = gamma.to(dev), beta.to(dev)
    ws = torch.empty(_lib.raw('mve_groupnorm_workspace_bytes')(B, HW, C, G), dtype=torch.uint8, device=dev)

    def gn(lo_t):
        out = torch.empty(B * HW, C, dtype=dtype, device=dev)
        h = hi.to(dev)
        l = lo_t.to(dev) if lo_t is not None else None
        _lib.call('mve_groupnorm_silu_pair', _dt(h), _lib.ptr(h), C, None, 0, B, HW, G, 1e-5, _lib.ptr(gamma_d), _lib.ptr(beta_d), 1,
                  _lib.ptr(out), _lib.ptr(ws), _lib.ptr(l), None, _lib.stream_ptr(dev))
        torch.cuda.synchronize()
        return out.double().cpu()
    plain = ops.groupnorm(hi.to(dev), B, HW, gamma_d, beta_d, G, 1e-5, True).double().cpu()
    assert torch.equal(gn(None), plain)
    assert torch.equal(gn(torch.zeros_like(lo)), plain)
    e_pair, e_plain = float((gn(lo) - ref).norm() / ref.norm()), float((plain - ref).norm() / ref.norm())
    print(f'{dtype}: GroupNorm+SiLU rel-L2 vs fp64 of the unrounded input: pair {e_pair:.2e}, hi alone {e_plain:.2e}')
    assert e_pair <= e_plain * 1.02
    # small tensors: the one-launch kernel reads pairs too (built in round 5, shipped in round 6 for <= 256 pixels per image) -- 16 x 16 and 8 x 8 pixels take it,
    # 32 x 32 stays on the three launches; one and two sources (the skip concatenation), B = 3 (no XCD map) and B = 8 (XCD map)
    for (Bs, HWs, C1s, C2s) in [(3, 256, 640, 0), (8, 1024, 320, 0), (3, 256, 1280, 1280), (2, 1024, 1280, 640), (3, 64, 1280, 0), (8, 64, 1280, 1280), (8, 256, 1280, 640)]:
        Cs = C1s + C2s
        xs = torch.randn(Bs * HWs, Cs, generator=g) * 2 + 0.3
        h1, l1 = _split_pair(xs[:, :C1s].contiguous(), dtype)
        h2, l2 = _split_pair(xs[:, C1s:].contiguous(), dtype) if C2s else (None, None)
        gam, bet = torch.rand(Cs, generator=g) + 0.5, torch.randn(Cs, generator=g) * 0.1
        refs = F.silu(F.group_norm(xs.view(Bs, HWs, Cs).permute(0, 2, 1).double(), G, gam.double(), bet.double(), 1e-5)).permute(0, 2, 1).reshape(Bs * HWs, Cs)
        wss = torch.empty(_lib.raw('mve_groupnorm_workspace_bytes')(Bs, HWs, Cs, G), dtype=torch.uint8, device=dev)
        gd, bd = gam.to(dev), bet.to(dev)

        def gns(with_lo, zero=False):
            o = torch.empty(Bs * HWs, Cs, dtype=dtype, device=dev)
            a, b_ = h1.to(dev), (h2.to(dev) if C2s else None)
            la = (torch.zeros_like(l1) if zero else l1).to(dev) if with_lo else None
            lb = ((torch.zeros_like(l2) if zero else l2).to(dev) if with_lo else None) if C2s else None
            _lib.call('mve_groupnorm_silu_pair', _dt(a), _lib.ptr(a), C1s, _lib.ptr(b_), C2s, Bs, HWs, G, 1e-5, _lib.ptr(gd), _lib.ptr(bd), 1,
                      _lib.ptr(o), _lib.ptr(wss), _lib.ptr(la), _lib.ptr(lb), _lib.stream_ptr(dev))
            torch.cuda.synchronize()
            return o.double().cpu()
        p0 = gns(False)
        if HWs <= 256:
            assert torch.equal(gns(True, zero=True), p0), (Bs, HWs, C1s, C2s)        # zero low halves: the plain kernel's bits (both on the one-launch kernel)
        else:                                                                          # 32 x 32: the pair stays on the three launches, the plain tensor takes one
            assert float((gns(True, zero=True) - p0).abs().max()) <= 4e-3 * float(p0.abs().max()), (Bs, HWs, C1s, C2s)
        e_pair, e_plain = float((gns(True) - refs).norm() / refs.norm()), float((p0 - refs).norm() / refs.norm())
        print(f'{dtype} GroupNorm B={Bs} HW={HWs} C={C1s}+{C2s}: pair {e_pair:.2e}, hi alone {e_plain:.2e}')
        assert e_pair <= e_plain * 1.02, (Bs, HWs, C1s, C2s)
        if Bs == 3:                                                                    # batch invariance of the pair path: image 1 alone == image 1 of the batch
            o = torch.empty(HWs, Cs, dtype=dtype, device=dev)
            sl = slice(HWs, 2 * HWs)
            a, b_ = h1[sl].contiguous().to(dev), (h2[sl].contiguous().to(dev) if C2s else None)
            la, lb = l1[sl].contiguous().to(dev), (l2[sl].contiguous().to(dev) if C2s else None)
            _lib.call('mve_groupnorm_silu_pair', _dt(a), _lib.ptr(a), C1s, _lib.ptr(b_), C2s, 1, HWs, G, 1e-5, _lib.ptr(gd), _lib.ptr(bd), 1,
                      _lib.ptr(o), _lib.ptr(wss), _lib.ptr(la), _lib.ptr(lb), _lib.stream_ptr(dev))
            torch.cuda.synchronize()
            assert torch.equal(o.double().cpu(), gns(True)[sl]), (Bs, HWs, C1s, C2s)
    # LayerNorm
    M = 4096
    x32 = torch.randn(M, C, generator=g) * 1.5
    hi, lo = _split_pair(x32, dtype)
    ref = F.layer_norm(x32.double(), (C,), gamma.double(), beta.double(), 1e-5)

    def ln(lo_t):
        out = torch.empty(M, C, dtype=dtype, device=dev)
        h = hi.to(dev)
        l = lo_t.to(dev) if lo_t is not None else None
        _lib.call('mve_layernorm_pair', _dt(h), _lib.ptr(h), C, _lib.ptr(out), C, M, C, _lib.ptr(gamma_d), _lib.ptr(beta_d), 1e-5,
                  _lib.ptr(l), _lib.stream_ptr(dev))
        torch.cuda.synchronize()
        return out.double().cpu()
    plain = ops.layernorm(hi.to(dev), gamma_d, beta_d).double().cpu()
    assert torch.equal(ln(None), plain)
    e_pair, e_plain = float((ln(lo) - ref).norm() / ref.norm()), float((plain - ref).norm() / ref.norm())
    print(f'{dtype}: LayerNorm rel-L2 vs fp64 of the unrounded input: pair {e_pair:.2e}, hi alone {e_plain:.2e}')
    assert e_pair <= e_plain * 1.02


@pytest.mark.gpu
def test_lo8_saturates_instead_of_overflowing_on_large_bf16_values(lib):
    """a bf16 stream value of magnitude >= 2^16 has a rounding remainder above E5M2's 57344 / 2^8; the packer clamps the scaled
    remainder (and zeroes a NaN one), so hi + lo8 stays finite and never lands further from the value than hi alone.  fp16 cannot reach the range."""
    from mvedit_amd import ops
    dtype = torch.bfloat16
    M, N, K = 256, 320, 64
    a = rnd((M, K), dtype, 1)
    w = rnd((N, K), dtype, 2, 3.0e5)                  # outputs of magnitude ~2e6 >> 2^16
    ref = a.double() @ w.double().t()
    hi, lo = ops.gemm(a.cuda(), w.cuda(), pair_out=True)
    got = hi.double().cpu() + _lo(lo)
    assert torch.isfinite(got).all() and float(ref.abs().max()) > 2.0 ** 18
    e_pair, e_hi = (got - ref).abs(), (hi.double().cpu() - ref).abs()
    assert bool((e_pair <= e_hi * (1 + 1e-6) + 1e-3).all()), 'the pair is never worse than its high half'
    assert float(e_pair.mean()) < float(e_hi.mean())
    # host-side converter: same clamp
    h2, l2 = ops.split_pair(ref.float(), dtype)
    assert torch.isfinite(ops.lo8_to_float(l2)).all()
