// Fragment of the executor's single translation unit (csrc/unet.hip includes it; not a stand-alone header): LPIPS(net='vgg') forward + backward plan.
#pragma once
#include "executor_builder.h"

namespace {

// LPIPS(net='vgg')(pred, target) and its gradient w.r.t. pred (lpips==0.1.4 as called from lib/models/losses/lpips_loss.py:8-42).
// Forward ops = [0, enc_end): both images of every pair go through VGG16 as one batch of 2B; nothing is released, the
// activations are the backward's inputs.  Backward ops = [enc_end, end): pred half only; every conv's dgrad is the forward conv
// kernel on the transposed / flipped weight packing.
int Builder::build_lpips(int B_, int H, int W, int io_dtype) {
    B = B_; dt = c.dtype;
    const int Bb = B_;
    pl = Plan();
    pl.B = Bb; pl.H = H; pl.W = W; pl.n_img = 1; pl.io_dtype = io_dtype;
    const int e = 2, d = dt, norm = c.lpips_normalize;
    ld_temb = 0; ld_kv = 0;
    MVE_CHECK(H % 16 == 0 && W % 16 == 0, MVE_ERR_ARG, "lpips: image size %dx%d must be divisible by 16 (four 2x2 poolings)", H, W);
    MVE_CHECK((size_t)2 * Bb * H * W * 64 < ((size_t)1 << 31), MVE_ERR_ARG, "lpips: batch %d at %dx%d overflows 32-bit activation indexing", Bb, H, W);
    Ref pred; pred.kind = Ref::SAMPLE;
    Ref targ; targ.kind = Ref::CTX;
    Ref loss; loss.kind = Ref::OUT;
    Ref shift = wt("shift"), scale = wt("scale"), zeros = wt("zeros");
    Ref x0 = ws((size_t)2 * Bb * H * W * 8 * e);
    op(OC_OTHER, 0, "scaling layer (nchw->nhwc)", [=](const Run& r) {
        return mve_lpips_scale(d, io_dtype, r.p(pred), r.p(targ), Bb, H, W, (const float*)r.p(shift), (const float*)r.p(scale), norm, r.p(x0), r.stream);
    });
    struct Act { Ref r; int C, h, w; };
    std::vector<Act> acts(13);          // post-ReLU output of every conv
    Ref cur = x0;
    int cin = 8, h = H, w = W;
    for (int k = 0; k < 5; ++k) {
        if (k > 0) {
            const int C = cin, hh = h, ww = w;
            Ref pooled = ws((size_t)2 * Bb * (h / 2) * (w / 2) * C * e);
            Ref in = cur;
            op(OC_OTHER, 0, "maxpool 2x2", [=](const Run& r) { return mve_maxpool2x2(d, r.p(in), 2 * Bb, hh, ww, C, r.p(pooled), r.stream); });
            cur = pooled; h /= 2; w /= 2;
        }
        rows_img = h * w;
        for (int i = VGG_BLK_FIRST[k]; i < VGG_BLK_FIRST[k + 1]; ++i) {
            const int C = VGG_COUT[i];
            const std::string en = "vgg." + std::to_string(i);
            Ref y = ws((size_t)2 * Bb * h * w * C * e);
            conv(cur, cin, 2 * Bb, h, w, 1, 0, wt(en + ".w"), C, y, wt(en + ".b"), Ref(), 0, Ref(), 0, "vgg conv");
            const size_t nel = (size_t)2 * Bb * h * w * C;
            op(OC_OTHER, 0, "relu", [=](const Run& r) { return mve_prelu(d, r.p(y), (const float*)r.p(zeros), C, r.p(y), nel, r.stream); });
            acts[i] = {y, C, h, w};
            cur = y; cin = C;
        }
        const int C = cin, hw = h * w;
        Ref lin = wt("lin." + std::to_string(k)), tap = cur;
        Ref scratch = ws(mve_lpips_layer_scratch_bytes(Bb, hw));
        const int acc = k > 0 ? 1 : 0;
        op(OC_OTHER, 0, "lpips layer distance", [=](const Run& r) {
            return mve_lpips_layer(d, r.p(tap), (const float*)r.p(lin), Bb, hw, C, acc, (float*)r.p(loss), r.p(scratch), r.stream);
        });
    }
    pl.enc_end = pl.ops.size();
    // ---- backward -----------------------------------------------------------------------------------------------------
    Ref gout; gout.kind = Ref::TIMESTEPS;          // d L / d loss[n], fp32 [B]
    Ref g;
    for (int k = 4; k >= 0; --k) {
        const int last = VGG_BLK_FIRST[k + 1] - 1, C = VGG_COUT[last], hw = h * w;
        Ref gf = ws((size_t)Bb * hw * C * e);
        Ref lin = wt("lin." + std::to_string(k)), tap = acts[last].r;
        op(OC_OTHER, 0, "lpips layer backward", [=](const Run& r) {
            return mve_lpips_layer_backward(d, r.p(tap), (const float*)r.p(lin), (const float*)r.p(gout), Bb, hw, C, r.p(gf), r.stream);
        });
        if (g.kind == Ref::NUL) g = gf;
        else {
            const size_t nel = (size_t)Bb * hw * C;
            Ref gg = g;
            op(OC_OTHER, 0, "grad += layer grad", [=](const Run& r) { return mve_axpy(d, r.p(gg), r.p(gf), 1.0f, r.p(gg), nel, r.stream); });
            rel(gf);
        }
        rows_img = hw;
        for (int i = last; i >= VGG_BLK_FIRST[k]; --i) {
            const int Co = VGG_COUT[i], Ci = i == 0 ? 8 : VGG_CIN[i];
            const size_t nel = (size_t)Bb * hw * Co;
            Ref gg = g, a = acts[i].r;
            op(OC_OTHER, 0, "relu backward", [=](const Run& r) { return mve_relu_backward(d, r.p(gg), r.p(a), nel, r.stream); });
            Ref gi = ws((size_t)Bb * hw * Ci * e);
            conv(g, Co, Bb, h, w, 1, 0, wt("vgg." + std::to_string(i) + ".wt"), Ci, gi, Ref(), Ref(), 0, Ref(), 0, "vgg conv dgrad");
            rel(g);
            g = gi;
        }
        if (k > 0) {
            const int Cp = VGG_COUT[VGG_BLK_FIRST[k] - 1], hh = 2 * h, ww = 2 * w;
            Ref gp = ws((size_t)Bb * hh * ww * Cp * e);
            Ref xin = acts[VGG_BLK_FIRST[k] - 1].r, gg = g;
            op(OC_OTHER, 0, "maxpool backward", [=](const Run& r) { return mve_maxpool2x2_backward(d, r.p(xin), r.p(gg), Bb, hh, ww, Cp, r.p(gp), r.stream); });
            rel(g);
            g = gp; h = hh; w = ww;
        }
    }
    {
        Ref gg = g;
        const int HH = H, WW = W;
        op(OC_OTHER, 0, "input gradient (nhwc->nchw)", [=](const Run& r) {
            return mve_lpips_input_grad(d, io_dtype, r.p(gg), Bb, HH, WW, (const float*)r.p(scale), norm, r.p(loss), r.stream);
        });
    }
    pl.ws_bytes = ar.peak + 256;
    if (!u.err.empty()) { mve_set_error("lpips plan: %s", u.err.c_str()); u.err.clear(); return MVE_ERR_STATE; }
    return MVE_OK;
}

}  // namespace
