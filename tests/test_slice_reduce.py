"""In-kernel K-slice reduction of the 256-row ping-pong tile (csrc/gemm_pp.hip: pp_reduce_slices; round 6, the small-M schedule).

A K-sliced GEMM / conv launch folds its slices inside the launch instead of leaving them to a k_splitk_reduce launch; launches of a rank that
holds few images thereby run on the ping-pong loop instead of the 128-row two-stage kernel.  Claims tested here, on the shapes one rank of an
8-GPU job sees at every level of the SD-1.5 UNet (8 images: 64 x 64 ... 8 x 8 latents) and on ragged ones:
  * bit-identical to partials + reducer (mve_gemm_red_tune(0)) and to the 128-row kernel (mve_gemm_tune(0)) -- same slices, same fold order;
  * within the fp16 / bf16 tolerance of an fp32 F.linear / F.conv2d reference;
  * repeatable (the counters return to zero: the 2nd, 3rd ... launch on the same stream gives the same bits), on side streams too;
  * batch invariant (an image alone == the same image inside the batch).
"""
import pytest
import torch
import torch.nn.functional as F

from test_unet_ops import TOL, check, conv_ref, rnd, to_nhwc, _split_pair, _lo, _pair_tol

pytestmark = pytest.mark.gpu


@pytest.fixture()
def red(lib):
    from mvedit_amd import _lib
    t = _lib.raw('mve_gemm_red_tune')
    old = t(-1)
    yield t
    t(old)


# (M, N, K, rows per image): the K-sliced linears of an 8-image forward (levels 1-3) + ragged / tiny / many-slice cases
LINEAR = [(8 * 1024, 640, 2560, 1024), (8 * 256, 1280, 1280, 256), (8 * 256, 1280, 5120, 256), (8 * 64, 1280, 1280, 64), (8 * 64, 1280, 5120, 64),
          (2 * 150, 1280, 5120, 150), (64, 1280, 1280, 64), (3 * 64 + 7, 1280, 2560, 64), (8 * 256, 1280, 2560, 256)]


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('M,N,K,rpi', LINEAR)
def test_linear_slices_folded_in_the_launch(lib, red, dtype, M, N, K, rpi):
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_gemm_tune')
    assert _lib.raw('mve_gemm_workspace_bytes')(M, N, K, rpi) > 0, 'the slice rule must cut this shape'
    a, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
    bias, res = rnd((N,), torch.float32, 3).cuda(), rnd((M, N), dtype, 4).cuda()
    red(2)                                           # ping-pong tile, slices folded in the launch
    o1 = ops.gemm(a, w, bias=bias, residual=res, rows_per_image=rpi)
    o1b = ops.gemm(a, w, bias=bias, residual=res, rows_per_image=rpi)
    red(1)                                           # 128-row kernel, slices folded in the launch
    o2 = ops.gemm(a, w, bias=bias, residual=res, rows_per_image=rpi)
    o2b = ops.gemm(a, w, bias=bias, residual=res, rows_per_image=rpi)
    assert torch.equal(o1, o2) and torch.equal(o2, o2b), '128-row fold != ping-pong fold'
    red(0)
    o0 = ops.gemm(a, w, bias=bias, residual=res, rows_per_image=rpi)
    old = tune(-1)
    try:
        tune(0)
        o128 = ops.gemm(a, w, bias=bias, residual=res, rows_per_image=rpi)
    finally:
        tune(old)
    assert torch.equal(o1, o0), 'in-kernel fold != partials + reducer'
    assert torch.equal(o1, o128), 'in-kernel fold != 128-row kernel + reducer'
    assert torch.equal(o1, o1b), 'second launch differs: counters not back at zero?'
    check('linear', o1, a.float().cpu() @ w.float().cpu().t() + bias.cpu() + res.float().cpu(), dtype, f'M={M} N={N} K={K}')
    red(3)
    rows = min(M, rpi)
    alone = ops.gemm(a[:rows], w, bias=bias, residual=res[:rows], rows_per_image=rpi)
    assert torch.equal(alone, o1[:rows]), 'batch invariance'


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_linear_slices_pair_geglu_and_streams(lib, red, dtype):
    """The residual-pair epilogue and GEGLU run through the in-kernel fold exactly as through the reducer; two streams use separate counters."""
    from mvedit_amd import ops
    M, N, K, rpi = 8 * 64, 1280, 5120, 64
    a, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
    bias = rnd((N,), torch.float32, 3).cuda()
    r32 = torch.randn(M, N, generator=torch.Generator().manual_seed(9)) * 3
    rh, rl = _split_pair(r32, dtype)
    outs = []
    for on in (2, 1, 0):
        red(on)
        hi, lo = ops.gemm(a, w, bias=bias, residual=rh.cuda(), residual_lo=rl.cuda(), rows_per_image=rpi, pair_out=True)
        g = ops.gemm(a, w, flags=ops.GEGLU, rows_per_image=rpi)
        outs.append((hi, lo, g))
    assert all(torch.equal(x, y) and torch.equal(x, z) for x, y, z in zip(*outs))
    hi, lo, _ = outs[0]
    ref = a.double().cpu() @ w.double().cpu().t() + bias.double().cpu() + (rh.double() + _lo(rl))
    assert float(((hi.double().cpu() + _lo(lo)) - ref).abs().max() / ref.abs().max()) < _pair_tol(dtype)
    red(3)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    got = []
    for s in (s1, s2, s1, s2):
        with torch.cuda.stream(s):
            got.append(ops.gemm(a, w, bias=bias, rows_per_image=rpi))
    torch.cuda.synchronize()
    assert all(torch.equal(got[0], g) for g in got[1:])


# (B, H, C1, C2, Cout): the K-sliced 3 x 3 convolutions of an 8-image forward + ragged cases
CONV = [(8, 8, 1280, 0, 1280), (8, 8, 1280, 1280, 1280), (8, 16, 640, 0, 1280), (8, 16, 1280, 640, 1280), (8, 32, 320, 0, 640), (8, 32, 640, 0, 640),
        (2, 10, 1280, 0, 1280), (1, 8, 1280, 0, 1280), (3, 16, 1280, 0, 1280)]


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('B,H,C1,C2,Cout', CONV)
def test_conv_slices_folded_in_the_launch(lib, red, dtype, B, H, C1, C2, Cout):
    from mvedit_amd import ops, _lib
    tune = _lib.raw('mve_gemm_tune')
    assert _lib.raw('mve_gemm_workspace_bytes')(B * H * H, Cout, 9 * (C1 + C2), H * H) > 0
    x1 = rnd((B, C1, H, H), dtype, 1)
    x2 = rnd((B, C2, H, H), dtype, 2) if C2 else None
    wt = rnd((Cout, C1 + C2, 3, 3), dtype, 3, (9 * (C1 + C2)) ** -0.5)
    bias = rnd((Cout,), torch.float32, 4)
    temb = rnd((B, Cout), torch.float32, 5).cuda()
    w_k, wflag = ops.pack_conv_weight(wt, True)
    a1, a2 = to_nhwc(x1).cuda(), (to_nhwc(x2).cuda() if C2 else None)
    kw = dict(x2=a2, bias=bias.cuda(), rowvec=temb, flags=wflag)
    red(2)
    o1 = ops.conv3x3(a1, w_k.cuda(), B, H, H, **kw)[0]
    o1b = ops.conv3x3(a1, w_k.cuda(), B, H, H, **kw)[0]
    red(1)
    o2 = ops.conv3x3(a1, w_k.cuda(), B, H, H, **kw)[0]
    assert torch.equal(o1, o2) and torch.equal(o2, ops.conv3x3(a1, w_k.cuda(), B, H, H, **kw)[0]), '128-row fold != ping-pong fold'
    red(0)
    o0 = ops.conv3x3(a1, w_k.cuda(), B, H, H, **kw)[0]
    old = tune(-1)
    try:
        tune(0)
        o128 = ops.conv3x3(a1, w_k.cuda(), B, H, H, **kw)[0]
    finally:
        tune(old)
    assert torch.equal(o1, o0) and torch.equal(o1, o128) and torch.equal(o1, o1b)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    ref = conv_ref(xin, wt, bias) + temb.cpu()[:, :, None, None]
    check('conv', o1, to_nhwc(ref), dtype, f'B={B} H={H} C={C1}+{C2}->{Cout}')
    red(3)
    alone = ops.conv3x3(a1[:H * H], w_k.cuda(), 1, H, H, x2=(a2[:H * H] if C2 else None), bias=bias.cuda(), rowvec=temb[:1], flags=wflag)[0]
    assert torch.equal(alone, o1[:H * H]), 'batch invariance'


@pytest.mark.parametrize('B,H,C', [(8, 8, 1280), (8, 16, 1280), (2, 16, 1280)])
def test_upsample_phases_and_pair_outputs_through_the_fold(lib, red, B, H, C):
    """Upsample2D as four 2 x 2 phase convs (grouped output rows) and a conv that leaves as a residual pair: fold == reducer, bit for bit."""
    from mvedit_amd import ops
    dtype = torch.float16
    x = to_nhwc(rnd((B, C, H, H), dtype, 1)).cuda()
    wt = rnd((C, C, 3, 3), dtype, 2, (9 * C) ** -0.5)
    bias = rnd((C,), torch.float32, 3).cuda()
    w_k, wflag = ops.pack_conv_weight(wt, True)
    r32 = torch.randn(B * H * H, C, generator=torch.Generator().manual_seed(9)) * 2
    rh, rl = _split_pair(r32, dtype)
    outs = []
    for on in (3, 1, 0):
        red(on)
        (hi, lo), _, _ = ops.conv3x3(x, w_k.cuda(), B, H, H, bias=bias, residual=rh.cuda(), residual_lo=rl.cuda(), flags=wflag, pair_out=True)
        row = [hi, lo]
        if ops.upsample_conv_phases_supported(C, C, B, H, H):
            w4 = ops.pack_upsample_phase_weights(wt.cuda(), dtype)
            ph, pl = ops.upsample_conv_phases(x, w4, B, H, H, bias=bias, pair_out=True)
            row += [ph, pl]
        outs.append(row)
    assert all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(*outs))
    hi, lo = outs[0][:2]
    ref = to_nhwc(conv_ref(rnd((B, C, H, H), dtype, 1), wt, bias.cpu())).double() + (rh.double() + _lo(rl))
    assert float(((hi.double().cpu() + _lo(lo)) - ref).abs().max() / ref.abs().max()) < _pair_tol(dtype)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_four_stage_ring_of_the_128_row_kernel_is_bit_identical(lib, red, dtype):
    """k_gemm_deep (csrc/gemm.hip): the 128-row kernel with three K tiles in flight for launches of a block or two per CU -- same tile, same MFMA
    order, same epilogue as the two-stage loop, so every bit must agree; dense, conv (slab-major and tap-major weights, concat, stride 2, fused
    shortcut), K slices that start inside a slab, K tails (K % 64 != 0), one-tile problems."""
    from mvedit_amd import ops, _lib
    deep, tune = _lib.raw('mve_gemm_deep_tune'), _lib.raw('mve_gemm_tune')
    old_d, old_t = deep(-1), tune(-1)
    red(0)
    try:
        tune(0)                                     # the 128-row kernel everywhere
        def both(fn):
            deep(0); a = fn(); deep(4096); b = fn()
            return a, b
        for (M, N, K, rpi, fl) in [(512, 1280, 1280, 64, 0), (512, 1280, 5120, 64, 0), (2048, 1280, 1280, 256, 0), (8, 1280, 1280, 0, 0), (77 * 8, 1280, 768, 0, 0),
                                   (300, 640, 328, 0, 0), (512, 2560, 320, 0, ops.GEGLU), (130, 128, 200, 0, 0), (64, 320, 64, 0, 0)]:
            a, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
            bias = rnd((N,), torch.float32, 3).cuda()
            res = None if fl else rnd((M, N), dtype, 4).cuda()
            x, y = both(lambda: ops.gemm(a, w, bias=bias, residual=res, flags=fl, rows_per_image=rpi))
            assert torch.equal(x, y), (M, N, K)
            ref = a.float().cpu() @ w.float().cpu().t() + bias.cpu()
            if fl:
                ref = ref[:, 0::2] * F.gelu(ref[:, 1::2])
            else:
                ref = ref + res.float().cpu()
            check('deep gemm', y, ref, dtype, f'M={M} N={N} K={K}')
        for (B, H, C1, C2, Cout, stride, chunk) in [(8, 8, 1280, 0, 1280, 1, True), (2, 16, 640, 320, 640, 1, True), (2, 16, 320, 0, 320, 2, True), (3, 9, 72, 0, 320, 1, False),
                                                    (1, 16, 320, 0, 128, 1, True)]:
            x1 = to_nhwc(rnd((B, C1, H, H), dtype, 1)).cuda()
            x2 = to_nhwc(rnd((B, C2, H, H), dtype, 2)).cuda() if C2 else None
            wt = rnd((Cout, C1 + C2, 3, 3), dtype, 3, (9 * (C1 + C2)) ** -0.5)
            w_k, wflag = ops.pack_conv_weight(wt, chunk)
            x, y = both(lambda: ops.conv3x3(x1, w_k.cuda(), B, H, H, x2=x2, stride=stride, flags=wflag)[0])
            assert torch.equal(x, y), (B, H, C1, C2, Cout, stride)
        B, H, C1, C3 = 2, 8, 1280, 1280
        x1, x3 = to_nhwc(rnd((B, C1, H, H), dtype, 1)).cuda(), to_nhwc(rnd((B, C3, H, H), dtype, 2)).cuda()
        w2 = torch.cat([ops.pack_conv_weight(rnd((C1, C1, 3, 3), dtype, 3, (9 * C1) ** -0.5), True)[0].reshape(C1, -1), rnd((C1, C3), dtype, 4, C3 ** -0.5)], 1).contiguous().cuda()
        x, y = both(lambda: ops.conv3x3_shortcut(x1, w2, B, H, H, x3))
        assert torch.equal(x, y)
    finally:
        deep(old_d); tune(old_t)


def test_weight_strip_major_block_order_is_a_pure_remap(lib, red):
    """GemmParams::w_major (launches whose activations are the smaller operand walk the row panels inside a weight strip): same tiles, same slices,
    another block order -- every bit must agree with the row-panel-major order (mve_gemm_deep_tune bit 30 turns it off), on the 128-row kernel, on
    the ping-pong tiles, K-sliced and not, with ragged tile counts."""
    from mvedit_amd import ops, _lib
    deep, tune = _lib.raw('mve_gemm_deep_tune'), _lib.raw('mve_gemm_tune')
    old_d, old_t = deep(-1), tune(-1)
    OFF = 1 << 30
    dtype = torch.float16
    try:
        for word in (old_t, 0):                                   # default dispatch / 128-row kernel only
            tune(word)
            for (M, N, K, rpi, fl) in [(512, 1280, 5120, 64, 0), (300, 1280, 1280, 150, 0), (512, 10240, 1280, 0, ops.GEGLU), (1000, 3840, 1280, 0, 0), (64, 1280, 1280, 64, 0)]:
                a, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
                bias = rnd((N,), torch.float32, 3).cuda()
                deep(0); x = ops.gemm(a, w, bias=bias, flags=fl, rows_per_image=rpi)
                deep(OFF); y = ops.gemm(a, w, bias=bias, flags=fl, rows_per_image=rpi)
                assert torch.equal(x, y), (word, M, N, K)
            for (B, H, C1, Cout) in [(8, 8, 1280, 1280), (3, 10, 640, 1280), (2, 16, 1280, 640)]:
                x1 = to_nhwc(rnd((B, C1, H, H), dtype, 1)).cuda()
                w_k, wflag = ops.pack_conv_weight(rnd((Cout, C1, 3, 3), dtype, 3, (9 * C1) ** -0.5), True)
                deep(0); x = ops.conv3x3(x1, w_k.cuda(), B, H, H, flags=wflag)[0]
                deep(OFF); y = ops.conv3x3(x1, w_k.cuda(), B, H, H, flags=wflag)[0]
                assert torch.equal(x, y), (word, B, H, C1, Cout)
    finally:
        deep(old_d); tune(old_t)
