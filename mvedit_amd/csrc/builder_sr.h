// Fragment of the executor's single translation unit (csrc/unet.hip includes it; not a stand-alone header): SRVGGNetCompact plan.
#pragma once
#include "executor_builder.h"

namespace {

// SRVGGNetCompact.forward (lib/models/decoders/image_space_ss.py:63-70): conv + PReLU stack at the input resolution, last conv to
// out_ch * r * r channels, PixelShuffle(r), plus the nearest-upsampled input.  H x W is the input size.
int Builder::build_sr(int B_, int H, int W, int io_dtype) {
    B = B_; dt = c.dtype;
    const int Bb = B_;
    pl = Plan();
    pl.B = Bb; pl.H = H; pl.W = W; pl.n_img = 1; pl.io_dtype = io_dtype;
    const int e = 2, d = dt, F = c.ch[0], r = c.sr_scale, last = 2 * (c.layers_per_block + 1);
    const int opad = (c.out_ch * r * r + 7) & ~7;
    ld_temb = 0; ld_kv = 0;
    MVE_CHECK((size_t)Bb * H * W * (size_t)(F > opad * 2 ? F : opad * 2) < ((size_t)1 << 31), MVE_ERR_ARG,
              "srvgg: batch %d at %dx%d overflows 32-bit activation indexing; enhance in smaller batches", Bb, H, W);
    const int M = Bb * H * W;
    rows_img = H * W;
    Ref src; src.kind = Ref::SAMPLE;
    Ref cur = ws((size_t)M * 8 * e);
    {
        const int in_ch = c.in_ch;
        Ref x_in = cur;
        op(OC_OTHER, 0, "nchw->nhwc", [=](const Run& rr) { return mve_nchw_to_nhwc(d, io_dtype, rr.p(src), Bb, in_ch, H, W, 8, rr.p(x_in), rr.stream); });
    }
    int cin = 8;
    for (int k = 0; k <= c.layers_per_block; ++k) {
        const std::string b = "body." + std::to_string(2 * k);
        Ref y = ws((size_t)M * F * e);
        conv(cur, cin, Bb, H, W, 1, 0, wt(b + ".w"), F, y, wt(b + ".b"), Ref(), 0, Ref(), 0, "conv");
        rel(cur);
        Ref a = wt("body." + std::to_string(2 * k + 1) + ".a");
        const size_t nel = (size_t)M * F;
        op(OC_OTHER, 0, "prelu", [=](const Run& rr) { return mve_prelu(d, rr.p(y), (const float*)rr.p(a), F, rr.p(y), nel, rr.stream); });
        cur = y; cin = F;
    }
    Ref o = ws((size_t)M * opad * 4);
    conv(cur, F, Bb, H, W, 1, 0, wt("body." + std::to_string(last) + ".w"), opad, o, wt("body." + std::to_string(last) + ".b"), Ref(), 0, Ref(),
         MVE_GEMM_OUT_F32, "conv (to r*r sub-pixels)");
    rel(cur);
    {
        Ref dst; dst.kind = Ref::OUT;
        const int oc = c.out_ch;
        live(o, "pixel shuffle");
        op(OC_OTHER, 0, "pixel shuffle + nearest-upsampled input", [=](const Run& rr) {
            return mve_pixel_shuffle_add(io_dtype, (const float*)rr.p(o), opad, rr.p(src), Bb, oc, H, W, r, rr.p(dst), rr.stream);
        });
    }
    pl.enc_end = pl.ops.size();
    pl.ws_bytes = ar.peak + 256;
    if (!u.err.empty()) { mve_set_error("srvgg plan: %s", u.err.c_str()); u.err.clear(); return MVE_ERR_STATE; }
    return MVE_OK;
}

}  // namespace
