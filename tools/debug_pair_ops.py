"""Round 5 bisect, op level: every form of 3 x 3 conv the residual-pair mode launches, on the 256-row PAIR tile (mve_gemm_tune word 1) and on the
128-row kernel (word 0), against an fp64 reference -- error size and WHERE the bad elements sit (row inside the tile, column, image border)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib, ops  # noqa: E402

tune = _lib.raw('mve_gemm_tune')
dt = torch.float16
g = torch.Generator().manual_seed(3)
rnd = lambda shape, s=1.0: (torch.randn(shape, generator=g) * s)


def split(x32):
    return ops.split_pair(x32, dt)


L = lambda lo8: ops.lo8_to_float(lo8.cpu()).double()


def report(name, got, ref, B, H, W):
    err = (got - ref).abs()
    rel = float((got - ref).norm() / ref.norm())
    bad = err > 1e-2 * ref.abs().max()
    msg = f'{name:58s} rel-L2 {rel:.3e}  max|err| {float(err.max()):.3e} (max|ref| {float(ref.abs().max()):.2f})  bad {int(bad.sum())}/{bad.numel()}'
    if bad.any():
        rows, cols = bad.nonzero(as_tuple=True)
        pix = rows % (H * W)
        y, x = pix // W, pix % W
        border = ((y == 0) | (y == H - 1) | (x == 0) | (x == W - 1)).float().mean()
        msg += (f'\n      bad rows mod 256: min {int((rows % 256).min())} max {int((rows % 256).max())} distinct {len(set((rows % 256).tolist()))};'
                f' row tiles hit {len(set((rows // 256).tolist()))}/{(B * H * W + 255) // 256}; cols min {int(cols.min())} max {int(cols.max())} distinct {len(set(cols.tolist()))};'
                f' on image border {float(border):.2f}; first (row, col) {int(rows[0])}, {int(cols[0])}: got {float(got[rows[0], cols[0]]):.4f} ref {float(ref[rows[0], cols[0]]):.4f}')
    print(msg, flush=True)


def nchw(x, B, H, W):
    return x.float().view(B, H, W, -1).permute(0, 3, 1, 2)


def nhwc(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


for (M, N, K) in [(65536, 320, 320), (65536, 320, 1280), (16384, 1280, 1280)]:       # dense GEMM, many tiles
    a, wt = rnd((M, K)).to(dt), rnd((N, K), K ** -0.5).to(dt)
    bias = rnd((N,))
    rh, rl = split(rnd((M, N), 2.0))
    ref = a.double() @ wt.double().t() + bias.double() + rh.double() + L(rl)
    for word in (1, 0):
        tune(word)
        hi, lo = ops.gemm(a.cuda(), wt.cuda(), bias=bias.cuda(), residual=rh.cuda(), residual_lo=rl.cuda(), pair_out=True)
        report(f'GEMM M={M} N={N} K={K} [{"256-row" if word else "128-row"}] bias + residual pair -> pair', (hi.double().cpu() + L(lo)), ref, 1, 256, M // 256)
    tune(256)
for (B, H, C, Cout) in [(4, 32, 320, 320), (16, 64, 320, 320)]:
    W = H
    x = rnd((B * H * W, C)).to(dt)
    w_oihw = rnd((Cout, C, 3, 3), (9 * C) ** -0.5).to(dt)
    w, fl = ops.pack_conv_weight(w_oihw)
    bias, rv = rnd((Cout,)), rnd((B, Cout))
    r32 = rnd((B * H * W, Cout), 2.0)
    rh, rl = split(r32)
    conv = F.conv2d(nchw(x, B, H, W).double(), w_oihw.double(), bias.double(), padding=1)
    ref_res = nhwc(conv) + rh.double() + L(rl)
    ref_rv = nhwc(conv + rv.double()[:, :, None, None]) + rh.double() + L(rl)
    ref_plain = nhwc(conv)
    # shortcut: conv(h2) + 1x1(x3) + b + b2
    C3 = 640
    x3 = rnd((B * H * W, C3)).to(dt)
    wsc = rnd((Cout, C3), C3 ** -0.5).to(dt)
    b2 = rnd((Cout,))
    wcat = torch.cat([w.reshape(Cout, -1), wsc], 1).contiguous()
    ref_sc = ref_plain + x3.double() @ wsc.double().t() + b2.double()
    # upsample conv (3 x 3 over the nearest-upsampled image), pair out
    Hs = H // 2
    xs = rnd((B * Hs * Hs, C)).to(dt)
    ref_up = nhwc(F.conv2d(F.interpolate(nchw(xs, B, Hs, Hs).double(), scale_factor=2.0, mode='nearest'), w_oihw.double(), bias.double(), padding=1))
    w4 = ops.pack_upsample_phase_weights(w_oihw.cuda())
    for word in (1, 0):
        tune(word)
        tag = f'B={B} {H}x{W} C={C}->{Cout} [{"256-row" if word else "128-row"}] '
        cu = lambda t: t.cuda()
        hi, lo = ops.conv3x3(cu(x), cu(w), B, H, W, bias=cu(bias), residual=cu(rh), residual_lo=cu(rl), flags=fl, splitk=False, pair_out=True)[0]
        report(tag + 'conv + bias + residual pair -> pair', (hi.double().cpu() + L(lo)), ref_res, B, H, W)
        hi, lo = ops.conv3x3(cu(x), cu(w), B, H, W, bias=cu(bias), rowvec=cu(rv), residual=cu(rh), residual_lo=cu(rl), flags=fl, splitk=False, pair_out=True)[0]
        report(tag + 'conv + bias + rowvec + residual pair -> pair', (hi.double().cpu() + L(lo)), ref_rv, B, H, W)
        hi, lo = ops.conv3x3(cu(x), cu(w), B, H, W, bias=cu(bias), flags=fl, splitk=False, pair_out=True)[0]
        report(tag + 'conv + bias -> pair (no residual)', (hi.double().cpu() + L(lo)), ref_plain, B, H, W)
        out = ops.conv3x3(cu(x), cu(w), B, H, W, bias=cu(bias), residual=cu(rh), flags=fl, splitk=False)[0]
        report(tag + 'conv + bias + 16-bit residual -> 16-bit (plain launch)', out.double().cpu(), nhwc(conv) + rh.double(), B, H, W)
        hi, lo = ops.conv3x3_shortcut(cu(x), cu(wcat), B, H, W, cu(x3), bias=cu(bias), bias2=cu(b2), splitk=False, pair_out=True)
        report(tag + 'conv + 1x1 shortcut + biases -> pair', (hi.double().cpu() + L(lo)), ref_sc, B, H, W)
        hi, lo = ops.conv3x3(cu(xs), cu(w), B, Hs, Hs, upsample=True, bias=cu(bias), flags=fl, splitk=False, pair_out=True)[0]
        report(tag + 'nearest-2x upsample + conv -> pair', (hi.double().cpu() + L(lo)), ref_up, B, H, W)
        if word == 1 and Cout % 320 == 0:
            hi, lo = ops.upsample_conv_phases(cu(xs), w4, B, Hs, Hs, bias=cu(bias), pair_out=True)
            report(tag + 'upsample as four phase convs -> pair', (hi.double().cpu() + L(lo)), ref_up, B, H, W)
    tune(256)
