// Fragment of the executor's single translation unit (csrc/unet.hip includes it; not a stand-alone header): UNet2DConditionModel / ControlNetModel plan (prologue, down path, ControlNet outputs or residuals, mid, up path, head).
#pragma once
#include "executor_builder.h"

namespace {

int Builder::build(int B_, int H, int W, int n_img, int has_res, int io_dtype, int res_nhwc) {
    B = B_; dt = c.dtype;
    const int Bb = B_;   // lambdas below must not capture `this`
    { const AttnOpts keep = pl.ao; pl = Plan(); pl.ao = keep; }
    pl.B = Bb; pl.H = H; pl.W = W; pl.n_img = n_img; pl.has_res = has_res; pl.io_dtype = io_dtype; pl.res_nhwc = res_nhwc;
    const int e = 2, n = c.n_levels, L = c.layers_per_block, T = c.temb_dim();
    const int d = dt;
    ld_temb = u.sum_temb; ld_kv = u.sum_kv;
    MVE_CHECK(Bb % n_img == 0, MVE_ERR_ARG, "unet: batch %d not divisible by num_cross_attn_imgs %d", Bb, n_img);
    MVE_CHECK((H % (1 << (n - 1))) == 0 && (W % (1 << (n - 1))) == 0, MVE_ERR_ARG,
              "unet: latent size %dx%d must be divisible by %d", H, W, 1 << (n - 1));
    for (int i = 0; i < n; ++i) {
        const int hd = c.ch[i] / c.heads[i];
        MVE_CHECK(!c.attn[i] || hd == 40 || hd == 64 || hd == 80 || hd == 160, MVE_ERR_ARG, "unet: unsupported head dim %d", hd);
    }
    // ---- prologue: layout conversion, time embedding, hoisted projections --------------------------
    const int M0 = Bb * H * W;
    Ref x_in = ws((size_t)M0 * 8 * e);
    {
        Ref src; src.kind = Ref::SAMPLE;
        const int in_ch = c.in_ch;
        op(OC_OTHER, 0, "nchw->nhwc", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(src), Bb, in_ch, H, W, 8, r.p(x_in), r.stream); });
    }
    Ref tsin = ws((size_t)Bb * c.ch[0] * e);
    {
        Ref tt; tt.kind = Ref::TIMESTEPS;
        const int dim = c.ch[0];
        op(OC_OTHER, 0, "timestep_embedding", [=](const Run& r) { return mve_timestep_embedding(d, (const float*)r.p(tt), Bb, dim, r.p(tsin), r.stream); });
    }
    Ref e1 = ws((size_t)Bb * T * e);
    gemm(tsin, c.ch[0], wt("time.w1"), c.ch[0], e1, T, Bb, T, c.ch[0], wt("time.b1"), Ref(), 0, 0, Ref(), 0, 0, "time_embedding.linear_1");
    rel(tsin);
    op(OC_OTHER, 0, "silu", [=](const Run& r) { return mve_silu(d, r.p(e1), r.p(e1), (size_t)Bb * T, r.stream); });
    Ref emb = ws((size_t)Bb * T * e);
    gemm(e1, T, wt("time.w2"), T, emb, T, Bb, T, T, wt("time.b2"), Ref(), 0, 0, Ref(), 0, 0, "time_embedding.linear_2");
    rel(e1);
    op(OC_OTHER, 0, "silu", [=](const Run& r) { return mve_silu(d, r.p(emb), r.p(emb), (size_t)Bb * T, r.stream); });
    tproj = ws((size_t)Bb * ld_temb * 4);
    gemm(emb, T, wt("temb_proj.w"), T, tproj, ld_temb, Bb, ld_temb, T, wt("temb_proj.b"), Ref(), 0, 0, Ref(), 0, MVE_GEMM_OUT_F32,
         "time_emb_proj (all resnets, one GEMM)");
    rel(emb);
    // encoder_hidden_states [Bb, Lc, ctx_dim]; under cross-image attention the text context is the mean of each group
    // (joint_attn.py:19-24).  K/V of every cross-attention layer in one GEMM.
    Ref ctx_src; ctx_src.kind = Ref::CTX;
    ctxB = Bb / n_img;
    Ref ctx_in = ctx_src, ctx_tmp;
    const int Lc = ctx_rows_per_img;
    const bool ctx_needs_copy = (io_dtype != dt) || n_img > 1;
    if (ctx_needs_copy) {
        ctx_tmp = ws((size_t)Bb * Lc * c.ctx_dim * e);
        // dtype conversion via the packing kernel semantics: reuse nchw->nhwc with H=W=1 treats [Bb*Lc, ctx] as NC11
        const int rows = Bb * Lc, cd = c.ctx_dim;
        op(OC_OTHER, 0, "ctx->dtype", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(ctx_src), rows, cd, 1, 1, cd, r.p(ctx_tmp), r.stream); });
        ctx_in = ctx_tmp;
        if (n_img > 1) {
            Ref cm = ws((size_t)ctxB * Lc * c.ctx_dim * e);
            const long long R = (long long)Lc * c.ctx_dim, total = (long long)ctxB * R;
            Ref in = ctx_tmp;
            op(OC_OTHER, 0, "ctx group mean", [=](const Run& r) {
                const unsigned grid = (unsigned)((total + 255) / 256);
                if (d == MVE_F16) k_group_mean<F16Tag><<<grid, 256, 0, r.stream>>>((const f16*)r.p(in), (f16*)r.p(cm), R, n_img, total);
                else k_group_mean<BF16Tag><<<grid, 256, 0, r.stream>>>((const bf16*)r.p(in), (bf16*)r.p(cm), R, n_img, total);
                return hipGetLastError() == hipSuccess ? MVE_OK : MVE_ERR_HIP;
            });
            ctx_in = cm;
        }
    }
    const AttnOpts ao = pl.ao;
    Lt = Lc - ao.ip_tokens;
    H0 = H;
    ref_off = 0;
    if (ao.ip_tokens > 0) {
        MVE_CHECK(Lt > 0, MVE_ERR_ARG, "unet: context of %d rows cannot hold %d ip tokens", Lc, ao.ip_tokens);
        MVE_CHECK(u.n_ip_loaded == 2 * u.n_xf_layers, MVE_ERR_STATE, "unet: IP-Adapter enabled but only %d of %d to_k_ip/to_v_ip weights loaded",
                  u.n_ip_loaded, 2 * u.n_xf_layers);
        // split [text | ip] rows of every item into two dense matrices (attention_processor.py:338-341)
        const size_t rowb = (size_t)c.ctx_dim * e;
        Ref ctx_text = ws((size_t)ctxB * Lt * rowb), ctx_ip = ws((size_t)ctxB * ao.ip_tokens * rowb);
        copy2d(ctx_text, Lt * rowb, ctx_in, Lc * rowb, Lt * rowb, ctxB, "ctx text rows");
        copy2d(ctx_ip, ao.ip_tokens * rowb, at(ctx_in, Lt * rowb), Lc * rowb, ao.ip_tokens * rowb, ctxB, "ctx ip rows");
        ipkv = ws((size_t)ctxB * ao.ip_tokens * ld_kv * e);
        gemm(ctx_ip, c.ctx_dim, wt("ip_kv.w"), c.ctx_dim, ipkv, ld_kv, ctxB * ao.ip_tokens, ld_kv, c.ctx_dim, Ref(), Ref(), 0, 0, Ref(), 0, 0,
             "ip-adapter K,V (all layers, one GEMM)");
        rel(ctx_ip);
        ctx_in = ctx_text;
    }
    if (ao.ref_mode) {
        MVE_CHECK(n_img == 1, MVE_ERR_ARG, "unet: reference attention and cross-image attention are exclusive (adapter3d_mixin.py:194)");
        MVE_CHECK(ao.ref_skip >= 0 && ao.ref_skip < Bb, MVE_ERR_ARG, "unet: ref_skip %d out of range", ao.ref_skip);
        if (ao.ref_mode == 2)
            MVE_CHECK(ao.ref_H > 0 && ao.ref_W > 0 && ao.ref_H % (1 << (n - 1)) == 0 && ao.ref_W % (1 << (n - 1)) == 0, MVE_ERR_ARG,
                      "unet: bad reference latent size %dx%d", ao.ref_H, ao.ref_W);
    }
    ctxkv = ws((size_t)ctxB * Lt * ld_kv * e);
    gemm(ctx_in, c.ctx_dim, wt("ctx_kv.w"), c.ctx_dim, ctxkv, ld_kv, ctxB * Lt, ld_kv, c.ctx_dim, Ref(), Ref(), 0, 0, Ref(), 0, 0,
         "cross-attention K,V (all layers, one GEMM)");
    // ---- conv_in + down path ---------------------------------------------------------------------------
    struct Skip { Ref r; int C, H, W; };
    std::vector<Skip> skips;
    Ref x = ws_stream((size_t)M0 * c.ch[0] * e);
    if (c.controlnet) {
        // controlnet_cond_embedding on the 8H x 8W conditioning image, added to conv_in(sample)
        const int Hc = 8 * H, Wc = 8 * W, cc = c.cond_ch;
        // cn_cond_repeat = R: Bc = Bb / R conditioning images, item b of the batch uses image b mod Bc (the two halves of a CFG batch share them)
        const int R = pl.ao.cn_cond_repeat > 1 ? pl.ao.cn_cond_repeat : 1;
        MVE_CHECK(Bb % R == 0, MVE_ERR_ARG, "controlnet: batch %d is not a multiple of the conditioning repeat %d", Bb, R);
        MVE_CHECK(R == 1 || !pl.ao.residual_pair, MVE_ERR_ARG, "controlnet: shared conditioning images are not combined with the residual-pair mode");
        const int Bc = Bb / R;
        Ref cimg = ws((size_t)Bc * Hc * Wc * 8 * e);
        {
            Ref src; src.kind = Ref::CNCOND;
            op(OC_OTHER, 0, "cond nchw->nhwc", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(src), Bc, cc, Hc, Wc, 8, r.p(cimg), r.stream); });
        }
        const std::string en = "controlnet_cond_embedding.";
        int hc = Hc, wc = Wc, ci = 8;
        Ref cur = cimg;
        auto emb_conv = [&](const std::string& nm, int co, int stride, bool act) {
            const int ho = (hc - 1) / stride + 1, wo = (wc - 1) / stride + 1;
            Ref y = ws((size_t)Bc * ho * wo * co * e);
            conv(cur, ci, Bc, hc, wc, stride, 0, wt(nm + ".w"), co, y, wt(nm + ".b"), Ref(), 0, Ref(), 0, "cond_embedding.conv");
            rel(cur);
            if (act) {
                const size_t nel = (size_t)Bc * ho * wo * co;
                op(OC_OTHER, 0, "silu", [=](const Run& r) { return mve_silu(d, r.p(y), r.p(y), nel, r.stream); });
            }
            cur = y; hc = ho; wc = wo; ci = co;
        };
        emb_conv(en + "conv_in", CN_EMB[0], 1, true);
        for (int k = 0; k < 6; ++k) emb_conv(en + "blocks." + std::to_string(k), CN_EMB[(k + 1) / 2], (k & 1) ? 2 : 1, true);
        emb_conv(en + "conv_out", c.ch[0], 1, false);
        MVE_CHECK(hc == H && wc == W, MVE_ERR_ARG, "controlnet: conditioning image must be 8x the latent size");
        for (int rep = 0; rep < R; ++rep) {      // one conv_in launch per group of Bc items: each adds the same embedding in its epilogue
            Ref xi = x_in, xo = x;
            xi.off += (size_t)rep * Bc * H * W * 8 * e;
            xo.off += (size_t)rep * Bc * H * W * c.ch[0] * e;
            conv(xi, 8, Bc, H, W, 1, 0, wt("conv_in.w"), c.ch[0], xo, wt("conv_in.b"), Ref(), 0, cur, 0, "conv_in + cond_embedding");
        }
        rel(cur);
    } else {
        conv(x_in, 8, Bb, H, W, 1, 0, wt("conv_in.w"), c.ch[0], x, wt("conv_in.b"), Ref(), 0, Ref(), 0, "conv_in");
    }
    rel(x_in);
    skips.push_back({x, c.ch[0], H, W});
    int h = H, w = W, cin = c.ch[0];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < L; ++j) {
            const std::string rn = "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
            Ref y = resnet(rn, x, cin, Ref(), 0, c.ch[i], h, w);
            cin = c.ch[i];
            if (c.attn[i]) {
                Ref z = transformer("down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), y, cin, c.heads[i], c.tlayers[i], h, w);
                rel(y);
                y = z;
            }
            x = y;
            skips.push_back({x, cin, h, w});
        }
        if (i + 1 < n) {
            const std::string dn = "down_blocks." + std::to_string(i) + ".downsamplers.0.conv";
            Ref y = ws_stream((size_t)Bb * (h / 2) * (w / 2) * cin * e);
            conv(x, cin, Bb, h, w, 2, 0, wt(dn + ".w"), cin, y, wt(dn + ".b"), Ref(), 0, Ref(), 0, "downsample");
            h /= 2; w /= 2;
            x = y;
            skips.push_back({x, cin, h, w});
        }
    }
    pl.enc_end = pl.ops.size();
    if (c.controlnet) {
        const int C = c.ch[n - 1];
        for (size_t i = 0; i < skips.size(); ++i)
            zero_conv(skips[i].r, skips[i].C, Bb * skips[i].H * skips[i].W, skips[i].H * skips[i].W, "controlnet_down_blocks." + std::to_string(i), (int)i);
        Ref y = resnet("mid_block.resnets.0", x, C, Ref(), 0, C, h, w);
        Ref z = transformer("mid_block.attentions.0", y, C, c.heads[n - 1], c.tlayers[n - 1], h, w);
        rel(y);
        Ref m = resnet("mid_block.resnets.1", z, C, Ref(), 0, C, h, w);
        rel(z);
        zero_conv(m, C, Bb * h * w, h * w, "controlnet_mid_block", (int)skips.size());
        pl.ws_bytes = ar.peak + 256;
        pl.ref_store_bytes = ref_off;
        if (!u.err.empty()) { mve_set_error("controlnet plan: %s", u.err.c_str()); u.err.clear(); return MVE_ERR_STATE; }
        return MVE_OK;
    }
    // ---- ControlNet residuals (diffusers.py:110-121 of the reference) -----------------------------------
    if (has_res) {
        for (size_t i = 0; i < skips.size(); ++i) {
            Skip& sk = skips[i];
            const size_t elems = (size_t)Bb * sk.H * sk.W * sk.C;
            Ref src; src.kind = Ref::DOWNRES; src.idx = (int)i;
            Ref sum = ws_stream(elems * e);
            Ref a = sk.r;
            const Ref a_lo = lo(a), sum_lo = lo(sum);
            const int C = sk.C, sh = sk.H, sw = sk.W;
            if (res_nhwc) {
                op(OC_OTHER, 0, "skip += controlnet residual", [=](const Run& r) { return mve_axpy_pair(d, r.p(a), r.p(a_lo), r.p(src), 1.0f, r.p(sum), r.p(sum_lo), elems, r.stream); });
            } else {
                Ref tmp = ws(elems * e);
                op(OC_OTHER, 0, "residual nchw->nhwc", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(src), Bb, C, sh, sw, C, r.p(tmp), r.stream); });
                op(OC_OTHER, 0, "skip += controlnet residual", [=](const Run& r) { return mve_axpy_pair(d, r.p(a), r.p(a_lo), r.p(tmp), 1.0f, r.p(sum), r.p(sum_lo), elems, r.stream); });
                rel(tmp);
            }
            sk.r = sum;    // the un-summed skip stays allocated: unet_enc state must survive unet_dec
        }
    }
    // ---- mid ------------------------------------------------------------------------------------------------
    {
        const int C = c.ch[n - 1];
        Ref y = resnet("mid_block.resnets.0", x, C, Ref(), 0, C, h, w);
        Ref z = transformer("mid_block.attentions.0", y, C, c.heads[n - 1], c.tlayers[n - 1], h, w);
        rel(y);
        Ref m = resnet("mid_block.resnets.1", z, C, Ref(), 0, C, h, w);
        rel(z);
        x = m;
        if (has_res) {
            const size_t elems = (size_t)Bb * h * w * C;
            Ref src; src.kind = Ref::MIDRES;
            Ref sum = ws_stream(elems * e);
            const Ref m_lo = lo(m), sum_lo = lo(sum);
            const int hh = h, ww = w;
            if (res_nhwc) {
                op(OC_OTHER, 0, "mid += controlnet residual", [=](const Run& r) { return mve_axpy_pair(d, r.p(m), r.p(m_lo), r.p(src), 1.0f, r.p(sum), r.p(sum_lo), elems, r.stream); });
            } else {
                Ref tmp = ws(elems * e);
                op(OC_OTHER, 0, "residual nchw->nhwc", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(src), Bb, C, hh, ww, C, r.p(tmp), r.stream); });
                op(OC_OTHER, 0, "mid += controlnet residual", [=](const Run& r) { return mve_axpy_pair(d, r.p(m), r.p(m_lo), r.p(tmp), 1.0f, r.p(sum), r.p(sum_lo), elems, r.stream); });
                rel(tmp);
            }
            rel(m);
            x = sum;
        }
    }
    // ---- up path ----------------------------------------------------------------------------------------------
    int cur = c.ch[n - 1];
    for (int i = 0; i < n; ++i) {
        const int lvl = n - 1 - i, cout = c.ch[lvl];
        for (int j = 0; j < L + 1; ++j) {
            Skip sk = skips.back();
            skips.pop_back();
            const std::string rn = "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
            Ref y = resnet(rn, x, cur, sk.r, sk.C, cout, h, w);
            rel(x);
            if (has_res) rel(sk.r);     // the summed copy; the original skip is enc state
            cur = cout;
            if (c.attn[lvl]) {
                Ref z = transformer("up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), y, cout, c.heads[lvl], c.tlayers[lvl], h, w);
                rel(y);
                y = z;
            }
            x = y;
        }
        if (i + 1 < n) {
            const std::string un = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
            Ref y = ws_stream((size_t)Bb * (2 * h) * (2 * w) * cout * e);
            upsample_conv(x, cout, Bb, h, w, un, y);
            rel(x);
            h *= 2; w *= 2;
            x = y;
        }
    }
    // ---- head ---------------------------------------------------------------------------------------------------
    Ref hn = ws((size_t)M0 * c.ch[0] * e);
    gn(x, c.ch[0], Ref(), 0, Bb, H * W, c.eps, wt("norm_out.g"), wt("norm_out.b"), 1, hn, "conv_norm_out+silu");
    rel(x);
    Ref o8 = ws((size_t)M0 * 8 * 4);
    conv(hn, c.ch[0], Bb, H, W, 1, 0, wt("conv_out.w"), 8, o8, wt("conv_out.b"), Ref(), 0, Ref(), MVE_GEMM_OUT_F32, "conv_out");
    rel(hn);
    {
        Ref dst; dst.kind = Ref::OUT;
        const int oc = c.out_ch;
        op(OC_OTHER, 0, "nhwc->nchw", [=](const Run& r) { return mve_nhwc_to_nchw(io_dtype, MVE_F32, r.p(o8), 8, Bb, oc, H, W, r.p(dst), r.stream); });
    }
    pl.ws_bytes = ar.peak + 256;
    pl.ref_store_bytes = ref_off;
    if (!u.err.empty()) { mve_set_error("unet plan: %s", u.err.c_str()); u.err.clear(); return MVE_ERR_STATE; }
    return MVE_OK;
}

}  // namespace
