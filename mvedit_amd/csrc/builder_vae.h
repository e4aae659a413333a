// Fragment of the executor's single translation unit (csrc/unet.hip includes it; not a stand-alone header): AutoencoderKL halves (mid-block attention on the GEMM kernel, decoder / encoder plans).
#pragma once
#include "executor_builder.h"

namespace {

// diffusers Attention of the VAE mid block (heads = 1, dim_head = C, residual_connection, bias everywhere, GroupNorm eps =
// resnet eps): x + to_out(softmax(q k^T / sqrt(C)) v).  Head dim C = 512 is outside the fused attention kernel's range, so the
// block runs on the GEMM kernel: scores (fp32) = q k^T, row softmax, P (V^T)^T with V^T produced directly by a GEMM whose "A"
// operand is the weight matrix.  to_v's bias is added after P.V (rows of P sum to one).  Images are processed one after the
// other through one [L, L] score buffer.
Ref Builder::vae_attention(const std::string& name, Ref x, int C, int H, int W) {
    const int L = H * W, M = B * L, e = 2, d = dt;
    rows_img = 0;
    Ref n0 = ws((size_t)M * C * e);
    gn(x, C, Ref(), 0, B, L, c.eps, wt(name + ".group_norm.g"), wt(name + ".group_norm.b"), 0, n0, "vae attention.group_norm");
    Ref qk = ws((size_t)M * 2 * C * e);
    gemm(n0, C, wt(name + ".qk.w"), C, qk, 2 * C, M, 2 * C, C, wt(name + ".qk.b"), Ref(), 0, 0, Ref(), 0, 0, "vae attention.to_q,to_k");
    Ref vT = ws((size_t)B * C * L * e);
    for (int b = 0; b < B; ++b)
        gemm(wt(name + ".v.w"), C, at(n0, (size_t)b * L * C * e), C, at(vT, (size_t)b * C * L * e), L, C, L, C, Ref(), Ref(), 0, 0, Ref(), 0, 0,
             "vae attention.to_v (transposed)");
    rel(n0);
    Ref S = ws((size_t)L * L * 4), P = ws((size_t)L * L * e), a = ws((size_t)M * C * e);
    const float scale = 1.0f / sqrtf((float)C);
    for (int b = 0; b < B; ++b) {
        Ref q = at(qk, (size_t)b * L * 2 * C * e);
        gemm(q, 2 * C, at(q, (size_t)C * e), 2 * C, S, L, L, L, C, Ref(), Ref(), 0, 0, Ref(), 0, MVE_GEMM_OUT_F32, "vae attention.q k^T", scale);
        live(S, "vae attention.softmax"); live(P, "vae attention.softmax");
        op(OC_ATTN, 0, "vae attention.softmax", [=](const Run& r) {
            return mve_softmax_rows(d, (const float*)r.p(S), (size_t)L, L, L, r.p(P), (size_t)L, r.stream);
        });
        gemm(P, L, at(vT, (size_t)b * C * L * e), L, at(a, (size_t)b * L * C * e), C, L, C, L, wt(name + ".v.b"), Ref(), 0, 0, Ref(), 0, 0,
             "vae attention.P V");
    }
    rel(S); rel(P); rel(qk); rel(vT);
    Ref out = ws((size_t)M * C * e);
    gemm(a, C, wt(name + ".o.w"), C, out, C, M, C, C, wt(name + ".o.b"), Ref(), 0, 0, x, C, 0, "vae attention.to_out+residual");
    rel(a);
    return out;
}

// AutoencoderKL half (diffusers 0.27.2 autoencoders/vae.py Decoder / Encoder, as called at lib/pipelines/mvedit_3d_pipeline.py:1260
// and :1441 of the reference).  H x W is the size of the half's INPUT (latent for the decoder, image for the encoder).
int Builder::build_vae(int B_, int H, int W, int io_dtype) {
    B = B_; dt = c.dtype;
    const int Bb = B_;
    pl = Plan();
    pl.B = Bb; pl.H = H; pl.W = W; pl.n_img = 1; pl.io_dtype = io_dtype;
    const int e = 2, n = c.n_levels, L = c.layers_per_block, d = dt, Cm = c.ch[n - 1];
    ld_temb = 0; ld_kv = 0;
    const int f = 1 << (n - 1);
    MVE_CHECK((H * W) % 8 == 0, MVE_ERR_ARG, "vae: input size %dx%d must have a multiple of 8 pixels", H, W);
    if (c.vae == 2) MVE_CHECK(H % f == 0 && W % f == 0 && ((H / f) * (W / f)) % 8 == 0, MVE_ERR_ARG, "vae: image size %dx%d must be divisible by %d", H, W, f);
    const int wide = n > 1 && c.ch[1] > c.ch[0] ? c.ch[1] : c.ch[0];      // widest tensor at image resolution
    MVE_CHECK((size_t)Bb * H * W * (c.vae == 1 ? (size_t)f * f : 1) * wide < ((size_t)1 << 31), MVE_ERR_ARG,
              "vae: batch %d at this size overflows 32-bit activation indexing; decode / encode in smaller batches", Bb);
    const int M0 = Bb * H * W;
    Ref x_in = ws((size_t)M0 * 8 * e);
    {
        Ref src; src.kind = Ref::SAMPLE;
        const int in_ch = c.in_ch;
        op(OC_OTHER, 0, "nchw->nhwc", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(src), Bb, in_ch, H, W, 8, r.p(x_in), r.stream); });
    }
    rows_img = H * W;
    int h = H, w = W;
    Ref x;
    if (c.vae == 1) {
        Ref z = ws((size_t)M0 * 8 * e);
        gemm(x_in, 8, wt("pq_conv.w"), 8, z, 8, M0, 8, 8, wt("pq_conv.b"), Ref(), 0, 0, Ref(), 0, 0, "post_quant_conv");
        rel(x_in);
        x = ws((size_t)M0 * Cm * e);
        conv(z, 8, Bb, H, W, 1, 0, wt("conv_in.w"), Cm, x, wt("conv_in.b"), Ref(), 0, Ref(), 0, "conv_in");
        rel(z);
    } else {
        x = ws((size_t)M0 * c.ch[0] * e);
        conv(x_in, 8, Bb, H, W, 1, 0, wt("conv_in.w"), c.ch[0], x, wt("conv_in.b"), Ref(), 0, Ref(), 0, "conv_in");
        rel(x_in);
        int cin = c.ch[0];
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < L; ++j) {
                Ref y = resnet("down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), x, cin, Ref(), 0, c.ch[i], h, w);
                rel(x);
                x = y; cin = c.ch[i];
            }
            if (i + 1 < n) {
                const std::string dn = "down_blocks." + std::to_string(i) + ".downsamplers.0.conv";
                Ref y = ws((size_t)Bb * (h / 2) * (w / 2) * cin * e);
                rows_img = (h / 2) * (w / 2);
                conv(x, cin, Bb, h, w, 2, 0, wt(dn + ".w"), cin, y, wt(dn + ".b"), Ref(), 0, Ref(), MVE_CONV_PAD_BR, "downsample (pad bottom/right)");
                rel(x);
                h /= 2; w /= 2;
                x = y;
            }
        }
    }
    {
        Ref y = resnet("mid_block.resnets.0", x, Cm, Ref(), 0, Cm, h, w);
        rel(x);
        Ref z = vae_attention("mid_block.attentions.0", y, Cm, h, w);
        rel(y);
        x = resnet("mid_block.resnets.1", z, Cm, Ref(), 0, Cm, h, w);
        rel(z);
    }
    int cur = Cm;
    if (c.vae == 1) {
        for (int i = 0; i < n; ++i) {
            const int cout = c.ch[n - 1 - i];
            for (int j = 0; j < L + 1; ++j) {
                Ref y = resnet("up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), x, cur, Ref(), 0, cout, h, w);
                rel(x);
                x = y; cur = cout;
            }
            if (i + 1 < n) {
                const std::string un = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
                Ref y = ws((size_t)Bb * (2 * h) * (2 * w) * cout * e);
                rows_img = 4 * h * w;
                upsample_conv(x, cout, Bb, h, w, un, y);
                rel(x);
                h *= 2; w *= 2;
                x = y;
            }
        }
    }
    // ---- head: GroupNorm + SiLU, conv_out (N padded to 8), for the encoder quant_conv on the 8 moments channels ------------
    const int Mo = Bb * h * w;
    rows_img = h * w;
    Ref hn = ws((size_t)Mo * cur * e);
    gn(x, cur, Ref(), 0, Bb, h * w, c.eps, wt("norm_out.g"), wt("norm_out.b"), 1, hn, "conv_norm_out+silu");
    rel(x);
    Ref o8 = ws((size_t)Mo * 8 * 4);
    if (c.vae == 1) {
        conv(hn, cur, Bb, h, w, 1, 0, wt("conv_out.w"), 8, o8, wt("conv_out.b"), Ref(), 0, Ref(), MVE_GEMM_OUT_F32, "conv_out");
        rel(hn);
    } else {
        Ref m8 = ws((size_t)Mo * 8 * e);
        conv(hn, cur, Bb, h, w, 1, 0, wt("conv_out.w"), 8, m8, wt("conv_out.b"), Ref(), 0, Ref(), 0, "conv_out");
        rel(hn);
        gemm(m8, 8, wt("pq_conv.w"), 8, o8, 8, Mo, 8, 8, wt("pq_conv.b"), Ref(), 0, 0, Ref(), 0, MVE_GEMM_OUT_F32, "quant_conv");
        rel(m8);
    }
    {
        Ref dst; dst.kind = Ref::OUT;
        const int oc = c.out_ch, ho = h, wo = w;
        op(OC_OTHER, 0, "nhwc->nchw", [=](const Run& r) { return mve_nhwc_to_nchw(io_dtype, MVE_F32, r.p(o8), 8, Bb, oc, ho, wo, r.p(dst), r.stream); });
    }
    pl.enc_end = pl.ops.size();
    pl.ws_bytes = ar.peak + 256;
    if (!u.err.empty()) { mve_set_error("vae plan: %s", u.err.c_str()); u.err.clear(); return MVE_ERR_STATE; }
    return MVE_OK;
}

}  // namespace
