// Fragment of the executor's single translation unit (csrc/unet.hip includes it; not a stand-alone header): the plan builder -- workspace arena, op recording, and the blocks shared by all networks (GEMM / conv / norm / attention ops, ResnetBlock2D, Transformer2DModel).
#pragma once
#include "executor_params.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// plan builder
// ---------------------------------------------------------------------------------------------------
struct Builder {
    Unet& u;
    Plan& pl;
    Arena ar;
    const Config& c;
    int B, dt;
    int ld_temb, ld_kv;
    Ref tproj, ctxkv, ipkv;  // hoisted projections
    int ctx_rows_per_img = 0, ctxB = 0;
    int Lt = 0;              // text rows of the context (ctx_rows_per_img - ip_tokens)
    int H0 = 0;              // latent height of this pass (reference-store geometry)
    size_t ref_off = 0;      // running offset into the reference K/V store

    Builder(Unet& u_, Plan& p) : u(u_), pl(p), c(u_.cfg) {}

    Ref ws(size_t bytes) { Ref r; r.kind = Ref::WS; r.off = ar.alloc(bytes); return r; }
    // residual_pair mode (AttnOpts::residual_pair): a tensor of the residual stream gets a companion of half its size for its 8-bit low half.  The
    // companion is found through the high half's workspace offset, so the ops below pick it up by themselves: an output with a companion is
    // written as a pair, a residual / normalised input with a companion is read as one.  Everything else -- every MFMA operand read -- sees the
    // high half alone, exactly the tensor of the plain mode.
    std::map<size_t, Ref> lo_of;
    Ref ws_stream(size_t bytes) {
        Ref r = ws(bytes);
        if (pl.ao.residual_pair) lo_of[r.off] = ws(bytes / 2);      // lo8: one byte per 16-bit element (common.h)
        return r;
    }
    Ref lo(const Ref& r) const {
        if (r.kind != Ref::WS) return Ref();
        auto it = lo_of.find(r.off);
        return it == lo_of.end() ? Ref() : it->second;
    }
    void rel(const Ref& r) {
        if (r.kind != Ref::WS) return;
        auto it = lo_of.find(r.off);
        if (it != lo_of.end()) { ar.release(it->second.off); lo_of.erase(it); }
        ar.release(r.off);
    }
    Ref wt(const std::string& n, size_t elem_off = 0) {
        auto it = u.params.find(n);
        Ref r;
        if (it == u.params.end()) { u.err = "missing packed parameter " + n; return r; }
        r.kind = Ref::WT;
        r.off = it->second.off + elem_off * (it->second.f32 ? 4 : 2);
        return r;
    }
    static Ref at(Ref r, size_t bytes) { r.off += bytes; return r; }
    // plan-time guard: every workspace operand of an op must lie inside a block that is allocated right now
    void live(const Ref& r, const char* what) {
        if (r.kind != Ref::WS) return;
        for (auto& blk : ar.b)
            if (!blk.free && r.off >= blk.off && r.off < blk.off + blk.size) return;
        if (u.err.empty()) u.err = std::string("operand used after release in ") + what;
    }
    void op(int cls, double flops, const char* what, std::function<int(const Run&)> fn) {
        pl.ops.push_back({cls, flops, what, std::move(fn)});
        pl.flops[cls] += flops;
    }

    int rows_img = 0;        // rows per image of the level being emitted (split-K granularity); 0: never split
    void gemm(Ref A, int lda, Ref W, int ldw, Ref out, int ldc, int M, int N, int K, Ref bias, Ref rowvec, int ldrv,
              int rpv, Ref res, int ldr, int flags, const char* what, float out_scale = 1.0f) {
        const int rimg = rows_img;
        const int d = dt;
        live(A, what); live(out, what); live(res, what); live(rowvec, what);
        const size_t skb = mve_gemm_workspace_bytes(M, N, K, rimg);
        Ref sk = skb ? ws(skb) : Ref();
        const Ref res_lo = lo(res), out_lo = lo(out);
        op(OC_LINEAR, 2.0 * M * N * K, what, [=](const Run& r) {
            return mve_gemm_pair(d, r.p(A), lda, r.p(W), ldw, r.p(out), ldc, M, N, K, (const float*)r.p(bias), (const float*)r.p(rowvec),
                                 ldrv, rpv, r.p(res), ldr, flags, out_scale, r.p(sk), skb, rimg, r.p(res_lo), r.p(out_lo), r.stream);
        });
        rel(sk);
    }
    // GEMM whose output rows go straight into a LayerNorm (BasicTransformerBlock norm1 / norm2 / norm3 behind proj_in / attn1.to_out / attn2.to_out):
    // mve_gemm_pair_ln normalises the rows in the producing tile's epilogue where the launch allows (N = 320 on the pair tile) and runs the LayerNorm
    // kernel behind the GEMM otherwise -- bit-identical either way, so the op list does not depend on the batch.  Counted as one linear op.
    void gemm_ln(Ref A, int lda, Ref W, int ldw, Ref out, int ldc, int M, int N, int K, Ref bias, Ref res, int ldr, Ref ln_out, Ref ln_g, Ref ln_b,
                 const char* what) {
        const int rimg = rows_img;
        const int d = dt;
        live(A, what); live(out, what); live(res, what); live(ln_out, what);
        const size_t skb = mve_gemm_workspace_bytes(M, N, K, rimg);
        Ref sk = skb ? ws(skb) : Ref();
        const Ref res_lo = lo(res), out_lo = lo(out);
        op(OC_LINEAR, 2.0 * M * N * K, what, [=](const Run& r) {
            return mve_gemm_pair_ln(d, r.p(A), lda, r.p(W), ldw, r.p(out), ldc, M, N, K, (const float*)r.p(bias), r.p(res), ldr, r.p(sk), skb, rimg,
                                    r.p(res_lo), r.p(out_lo), r.p(ln_out), N, (const float*)r.p(ln_g), (const float*)r.p(ln_b), 1e-5f, r.stream);
        });
        rel(sk);
    }
    void conv(Ref x, int C1, int Bn, int H, int W, int stride, int ups, Ref Wt, int Cout, Ref out, Ref bias, Ref rowvec,
              int ldrv, Ref res, int flags, const char* what) {
        const int d = dt;
        const int Hv = ups ? 2 * H : H, Wv = ups ? 2 * W : W;
        const int Ho = (Hv - 1) / stride + 1, Wo = (Wv - 1) / stride + 1;
        live(x, what); live(out, what); live(res, what); live(rowvec, what);
        const int fl = flags | (C1 % 64 == 0 ? MVE_CONV_W_CHUNK64 : 0);   // must mirror load_param's packing rule
        const size_t skb = mve_gemm_workspace_bytes(Bn * Ho * Wo, Cout, 9 * C1, Ho * Wo);
        Ref sk = skb ? ws(skb) : Ref();
        const Ref res_lo = lo(res), out_lo = (flags & MVE_GEMM_OUT_F32) ? Ref() : lo(out);
        op(OC_CONV, 2.0 * Bn * Ho * Wo * (double)Cout * 9 * C1, what, [=](const Run& r) {
            return mve_conv3x3_pair(d, r.p(x), C1, nullptr, 0, Bn, H, W, stride, ups, r.p(Wt), Cout, r.p(out), Cout,
                                    (const float*)r.p(bias), (const float*)r.p(rowvec), ldrv, r.p(res), Cout, fl, 1.0f, r.p(sk), skb, r.p(res_lo),
                                    r.p(out_lo), r.stream);
        });
        rel(sk);
    }
    // Upsample2D: nearest 2x + conv 3x3.  Default: four 2 x 2 phase convs over the source (mve_upsample_conv_phases: 4 / 9 of the flops, one extra
    // rounding on the summed weights) wherever the shape allows; the 3 x 3 conv over the virtual upsampled image otherwise and with
    // MVE_UPSAMPLE_PHASES=0.  Round 5: also in residual_pair mode (now the default mode) -- the summed-weight rounding costs the end-to-end error
    // 8.5e-4 -> 8.7e-4 against fp32 (tests/rounding_budget_experiment.py --phase), inside north_star's 1e-3; the pair output needs the 320-wide
    // tile (C % 320 == 0: every UNet width).  The choice depends on the shape and the mode only, except below 64 source pixels per launch (tiny
    // test sizes at batch 1), where the 3 x 3 form runs.
    static bool upsample_phases_on() {
        static int on = -1;
        if (on < 0) { const char* e = getenv("MVE_UPSAMPLE_PHASES"); on = e ? (atoi(e) != 0) : 1; }
        return on != 0;
    }
    void upsample_conv(Ref x, int C, int Bn, int H, int W, const std::string& slot, Ref out) {
        const bool have4 = u.params.count(slot + ".w4") != 0;
        if (!have4 || !upsample_phases_on() || (pl.ao.residual_pair && C % 320 != 0) || !mve_upsample_conv_phases_supported(C, C, Bn, H, W)) {
            conv(x, C, Bn, H, W, 1, 1, wt(slot + ".w"), C, out, wt(slot + ".b"), Ref(), 0, Ref(), 0, "upsample+conv");
            return;
        }
        const int d = dt;
        const char* what = "upsample+conv (4 phases)";
        live(x, what); live(out, what);
        const size_t skb = mve_upsample_conv_phases_workspace_bytes(C, C, Bn, H, W);
        Ref sk = skb ? ws(skb) : Ref();
        const Ref W4 = wt(slot + ".w4"), bias = wt(slot + ".b"), out_lo = lo(out);
        // (flops: the op's ALGORITHMIC count -- the 3 x 3 conv over the upsampled image the reference computes, SURVEY.md 8(d) -- of which this form
        // executes 4 / 9; bench.py reports both)
        op(OC_CONV, 2.0 * Bn * (4.0 * H * W) * (double)C * 9 * C, what, [=](const Run& r) {
            return mve_upsample_conv_phases(d, r.p(x), C, Bn, H, W, r.p(W4), C, r.p(out), (const float*)r.p(bias), 0, r.p(sk), skb, r.p(out_lo), r.stream);
        });
        rel(sk);
    }
    void gn(Ref x1, int C1, Ref x2, int C2, int Bn, int HW, float eps, Ref g, Ref b, int silu, Ref out, const char* what) {
        const int d = dt, G = c.groups;
        const size_t wsb = mve_groupnorm_workspace_bytes(Bn, HW, C1 + C2, G);
        Ref scratch = ws(wsb);
        live(x1, what); live(x2, what); live(out, what);
        const Ref x1_lo = lo(x1), x2_lo = lo(x2);
        op(OC_NORM, 0, what, [=](const Run& r) {
            return mve_groupnorm_silu_pair(d, r.p(x1), C1, r.p(x2), C2, Bn, HW, G, eps, (const float*)r.p(g), (const float*)r.p(b), silu,
                                           r.p(out), r.p(scratch), r.p(x1_lo), r.p(x2_lo), r.stream);
        });
        rel(scratch);
    }
    void ln(Ref x, Ref y, int M, int C, Ref g, Ref b) {
        const int d = dt;
        live(x, "layernorm"); live(y, "layernorm");
        const Ref x_lo = lo(x);
        op(OC_NORM, 0, "layernorm", [=](const Run& r) {
            return mve_layernorm_pair(d, r.p(x), C, r.p(y), C, M, C, (const float*)r.p(g), (const float*)r.p(b), 1e-5f, r.p(x_lo), r.stream);
        });
    }
    // Q as mve_attention_prescaled wants it: the output of a GEMM over a to_q weight slot that carries the softmax scale.  The only way to get one
    // is prescaled(): a caller that projects Q with unfolded weights cannot reach attn() by accident.
    struct PrescaledQ { Ref q; };
    PrescaledQ prescaled(Ref q, const std::string& weight_slot, int hd) {
        auto it = u.params.find(weight_slot);
        const float want = 1.4426950408889634f / sqrtf((float)hd);
        if (it == u.params.end() || !(fabsf(it->second.q_fold - want) <= 1e-6f * want))
            u.err = "attention: Q is not produced by a to_q slot folded with head_dim^-1/2 * log2(e) (" + weight_slot + ")";
        return PrescaledQ{q};
    }
    void attn(PrescaledQ pq, int ldq, Ref k, int ldk, Ref v, int ldv, Ref o, int ldo, int Bn, int Lq, int Lk, int heads, int hd,
              Ref k2 = Ref(), int ldk2 = 0, Ref v2 = Ref(), int ldv2 = 0, int Lk2 = 0, const char* what = "attention") {
        const int d = dt;
        const Ref q = pq.q;
        if (Bn <= 0) return;
        live(q, what); live(k, what); live(v, what); live(o, what);
        op(OC_ATTN, 4.0 * Bn * heads * (double)Lq * (Lk + Lk2) * hd, what, [=](const Run& r) {
            // Q comes out of to_q already multiplied by hd^-1/2 * log2(e) (executor_params.h: q_fold)
            return mve_attention_prescaled(d, r.p(q), ldq, r.p(k), ldk, r.p(v), ldv, r.p(k2), ldk2, r.p(v2), ldv2, r.p(o), ldo, Bn, Lq, Lk, Lk2,
                                           heads, hd, r.stream);
        });
    }
    // device-to-device 2-D copy (rows x width bytes) between pitched buffers
    void copy2d(Ref dst, size_t dpitch, Ref src, size_t spitch, size_t width, size_t rows, const char* what) {
        if (!rows || !width) return;
        live(dst, what); live(src, what);
        op(OC_OTHER, 0, what, [=](const Run& r) {
            return hipMemcpy2DAsync(r.p(dst), dpitch, r.p(src), spitch, width, rows, hipMemcpyDeviceToDevice, r.stream) == hipSuccess
                       ? MVE_OK : MVE_ERR_HIP;
        });
    }

    // ControlNet output k: out_k (+)= conditioning_scale * (W x + b), a 1x1 "zero convolution" (diffusers ControlNetModel
    // controlnet_down_blocks / controlnet_mid_block, then the `* conditioning_scale` and MultiControlNetModel's running sum)
    void zero_conv(Ref x, int C, int M, int hw, const std::string& name, int out_idx) {
        const int d = dt;
        Ref W = wt(name + ".w"), bias = wt(name + ".b");
        Ref out; out.kind = Ref::CNOUT; out.idx = out_idx;
        live(x, "controlnet zero conv");
        const size_t skb = mve_gemm_workspace_bytes(M, C, C, hw);
        Ref sk = skb ? ws(skb) : Ref();
        op(OC_LINEAR, 2.0 * M * (double)C * C, "controlnet zero conv", [=](const Run& r) {
            void* o = r.p(out);
            return mve_gemm(d, r.p(x), C, r.p(W), C, o, C, M, C, C, (const float*)r.p(bias), nullptr, 0, 0, r.cn_accum ? o : nullptr, C,
                            MVE_GEMM_RES_AFTER_SCALE, r.cn_scale, r.p(sk), skb, hw, r.stream);
        });
        rel(sk);
    }

    // ResnetBlock2D.  x [M,C1] (+ skip [M,C2]) -> new buffer [M,Cout]
    Ref resnet(const std::string& name, Ref x, int C1, Ref skip, int C2, int Cout, int H, int W) {
        const int M = B * H * W, Cin = C1 + C2, e = 2;
        rows_img = H * W;
        Ref h0 = ws((size_t)M * Cin * e);
        gn(x, C1, skip, C2, B, H * W, c.eps, wt(name + ".norm1.g"), wt(name + ".norm1.b"), 1, h0, "resnet.norm1+silu");
        Ref h1 = ws((size_t)M * Cout * e);
        Ref tv = c.vae ? Ref() : at(tproj, (size_t)u.temb_off[name] * 4);      // the VAE's resnets have no time embedding
        conv(h0, Cin, B, H, W, 1, 0, wt(name + ".conv1.w"), Cout, h1, wt(name + ".conv1.b"), tv, ld_temb, Ref(), 0, "resnet.conv1");
        rel(h0);
        Ref h2 = ws((size_t)M * Cout * e);
        gn(h1, Cout, Ref(), 0, B, H * W, c.eps, wt(name + ".norm2.g"), wt(name + ".norm2.b"), 1, h2, "resnet.norm2+silu");
        rel(h1);
        if (Cin != Cout && u.fuse_sc) {
            // conv2 and the 1x1 conv_shortcut over [x | skip] share one K loop (mve_conv3x3_shortcut); no shortcut tensor exists
            Ref out = ws_stream((size_t)M * Cout * e);
            const Ref out_lo = lo(out);
            const int d = dt, Bn = B;
            Ref Wt = wt(name + ".conv2.w"), b2 = wt(name + ".conv2.b"), bs = wt(name + ".sc.b");
            live(h2, "resnet.conv2+shortcut"); live(x, "resnet.conv2+shortcut"); live(skip, "resnet.conv2+shortcut");
            const size_t skb = mve_gemm_workspace_bytes(M, Cout, 9 * Cout + Cin, H * W);
            Ref sk = skb ? ws(skb) : Ref();
            op(OC_CONV, 2.0 * M * (double)Cout * (9 * Cout + Cin), "resnet.conv2+shortcut", [=](const Run& r) {
                return mve_conv3x3_shortcut_pair(d, r.p(h2), Cout, r.p(x), C1, r.p(skip), C2, Bn, H, W, r.p(Wt), Cout, r.p(out), Cout,
                                                 (const float*)r.p(b2), (const float*)r.p(bs), 0, 1.0f, r.p(sk), skb, r.p(out_lo), r.stream);
            });
            rel(sk);
            rel(h2);
            return out;
        }
        Ref res = x, sc;
        if (Cin != Cout) {
            sc = ws((size_t)M * Cout * e);
            gemm(x, C1, wt(name + ".sc.w"), Cin, sc, Cout, M, Cout, C1, wt(name + ".sc.b"), Ref(), 0, 0, Ref(), 0, 0, "resnet.shortcut");
            if (C2) gemm(skip, C2, wt(name + ".sc.w", C1), Cin, sc, Cout, M, Cout, C2, Ref(), Ref(), 0, 0, sc, Cout, 0, "resnet.shortcut(skip)");
            res = sc;
        }
        Ref out = ws_stream((size_t)M * Cout * e);
        conv(h2, Cout, B, H, W, 1, 0, wt(name + ".conv2.w"), Cout, out, wt(name + ".conv2.b"), Ref(), 0, res, 0, "resnet.conv2");
        rel(h2);
        rel(sc);
        return out;
    }

    // Transformer2DModel.  x [M,C] -> new buffer [M,C]
    Ref transformer(const std::string& name, Ref x, int C, int heads, int layers, int H, int W) {
        const int M = B * H * W, e = 2, hd = C / heads;
        rows_img = H * W;
        const int nb = B / pl.n_img, L = H * W * pl.n_img;     // cross-image attention: [n*b, L, C] seen as [b, n*L, C]
        Ref n0 = ws((size_t)M * C * e);
        gn(x, C, Ref(), 0, B, H * W, 1e-6f, wt(name + ".norm.g"), wt(name + ".norm.b"), 0, n0, "transformer.norm");
        Ref h = ws_stream((size_t)M * C * e);
        // (round 6) a GEMM that writes the residual stream also writes the LayerNorm of its rows (gemm_ln): the row's next reader never fetches it back
        const bool fuse_ln = layers >= 1;      // (either stream mode: the op list does not depend on it; without a pair the LayerNorm kernel runs behind the GEMM)
        Ref n1_first;
        if (fuse_ln) {
            const std::string b0 = name + ".transformer_blocks.0";
            n1_first = ws((size_t)M * C * e);
            gemm_ln(n0, C, wt(name + ".proj_in.w"), C, h, C, M, C, C, wt(name + ".proj_in.b"), Ref(), 0, n1_first, wt(b0 + ".norm1.g"), wt(b0 + ".norm1.b"),
                    "transformer.proj_in+norm1");
        } else {
            gemm(n0, C, wt(name + ".proj_in.w"), C, h, C, M, C, C, wt(name + ".proj_in.b"), Ref(), 0, 0, Ref(), 0, 0, "transformer.proj_in");
        }
        rel(n0);
        for (int k = 0; k < layers; ++k) {
            const std::string b = name + ".transformer_blocks." + std::to_string(k);
            // self attention
            Ref n1;
            if (fuse_ln && k == 0) n1 = n1_first;
            else {
                n1 = ws((size_t)M * C * e);
                ln(h, n1, M, C, wt(b + ".norm1.g"), wt(b + ".norm1.b"));
            }
            Ref qkv = ws((size_t)M * 3 * C * e);
            gemm(n1, C, wt(b + ".qkv.w"), C, qkv, 3 * C, M, 3 * C, C, Ref(), Ref(), 0, 0, Ref(), 0, 0, "attn1.qkv");
            rel(n1);
            Ref a = ws((size_t)M * C * e);
            const AttnOpts& ao = pl.ao;
            if (ao.ref_mode == 0) {
                attn(prescaled(qkv, b + ".qkv.w", hd), 3 * C, at(qkv, (size_t)C * e), 3 * C, at(qkv, (size_t)2 * C * e), 3 * C, a, C, nb, L, L, heads, hd);
            } else {
                // reference attention: the store holds, per self-attention layer in execution order, the keys|values
                // [B - ref_skip][Lref][2C] of the pass that ran in 'w' mode (to_k / to_v act per token, so K(cat[x, ref]) =
                // cat[K(x), K(ref)] and the projected rows can be stored instead of the layer input)
                const int div = H0 / H;
                const int Lref = ao.ref_mode == 1 ? L : (ao.ref_H / div) * (ao.ref_W / div);
                const int skip = ao.ref_skip, nr = nb - skip;
                Ref st; st.kind = Ref::REFSTORE; st.off = ref_off;
                ref_off += (size_t)nr * Lref * 2 * C * e;
                if (ao.ref_mode == 1) {
                    attn(prescaled(qkv, b + ".qkv.w", hd), 3 * C, at(qkv, (size_t)C * e), 3 * C, at(qkv, (size_t)2 * C * e), 3 * C, a, C, nb, L, L, heads, hd);
                    copy2d(st, (size_t)2 * C * e, at(qkv, ((size_t)skip * L * 3 * C + C) * e), (size_t)3 * C * e, (size_t)2 * C * e,
                           (size_t)nr * L, "reference K,V -> store");
                } else {
                    attn(prescaled(qkv, b + ".qkv.w", hd), 3 * C, at(qkv, (size_t)C * e), 3 * C, at(qkv, (size_t)2 * C * e), 3 * C, a, C, skip, L, L, heads, hd);
                    const size_t o = (size_t)skip * L * 3 * C * e;
                    attn(prescaled(at(qkv, o), b + ".qkv.w", hd), 3 * C, at(qkv, o + (size_t)C * e), 3 * C, at(qkv, o + (size_t)2 * C * e), 3 * C,
                         at(a, (size_t)skip * L * C * e), C, nr, L, L, heads, hd, st, 2 * C, at(st, (size_t)C * e), 2 * C, Lref,
                         "attention (+reference tokens)");
                }
            }
            rel(qkv);
            Ref h2 = ws_stream((size_t)M * C * e);
            Ref n2 = ws((size_t)M * C * e);
            if (fuse_ln) {
                gemm_ln(a, C, wt(b + ".o1.w"), C, h2, C, M, C, C, wt(b + ".o1.b"), h, C, n2, wt(b + ".norm2.g"), wt(b + ".norm2.b"), "attn1.to_out+residual+norm2");
                rel(a); rel(h); h = h2;
            } else {
                gemm(a, C, wt(b + ".o1.w"), C, h2, C, M, C, C, wt(b + ".o1.b"), Ref(), 0, 0, h, C, 0, "attn1.to_out+residual");
                rel(a); rel(h); h = h2;
                // cross attention (K/V hoisted)
                ln(h, n2, M, C, wt(b + ".norm2.g"), wt(b + ".norm2.b"));
            }
            Ref q = ws((size_t)M * C * e);
            gemm(n2, C, wt(b + ".q2.w"), C, q, C, M, C, C, Ref(), Ref(), 0, 0, Ref(), 0, 0, "attn2.to_q");
            rel(n2);
            Ref a2 = ws((size_t)M * C * e);
            const size_t ko = (size_t)u.kv_off[b] * e;
            attn(prescaled(q, b + ".q2.w", hd), C, at(ctxkv, ko), ld_kv, at(ctxkv, ko + (size_t)C * e), ld_kv, a2, C, nb, L, Lt, heads, hd);
            if (ao.ip_tokens > 0) {     // hidden_states + scale * SDPA(q, to_k_ip(ip), to_v_ip(ip))  (attention_processor.py:366-383)
                Ref aip = ws((size_t)M * C * e);
                attn(prescaled(q, b + ".q2.w", hd), C, at(ipkv, ko), ld_kv, at(ipkv, ko + (size_t)C * e), ld_kv, aip, C, nb, L, ao.ip_tokens, heads, hd, Ref(), 0, Ref(), 0,
                     0, "attention (ip tokens)");
                const int d = dt;
                const float sc = ao.ip_scale;
                const size_t nel = (size_t)M * C;
                op(OC_OTHER, 0, "attn2 += scale * ip", [=](const Run& r) { return mve_axpy(d, r.p(a2), r.p(aip), sc, r.p(a2), nel, r.stream); });
                rel(aip);
            }
            rel(q);
            Ref h3 = ws_stream((size_t)M * C * e);
            Ref n3 = ws((size_t)M * C * e);
            if (fuse_ln) {
                gemm_ln(a2, C, wt(b + ".o2.w"), C, h3, C, M, C, C, wt(b + ".o2.b"), h, C, n3, wt(b + ".norm3.g"), wt(b + ".norm3.b"), "attn2.to_out+residual+norm3");
                rel(a2); rel(h); h = h3;
            } else {
                gemm(a2, C, wt(b + ".o2.w"), C, h3, C, M, C, C, wt(b + ".o2.b"), Ref(), 0, 0, h, C, 0, "attn2.to_out+residual");
                rel(a2); rel(h); h = h3;
                // feed forward (GEGLU fused in the first GEMM's epilogue)
                ln(h, n3, M, C, wt(b + ".norm3.g"), wt(b + ".norm3.b"));
            }
            Ref f = ws((size_t)M * 4 * C * e);
            gemm(n3, C, wt(b + ".ff1.w"), C, f, 4 * C, M, 8 * C, C, wt(b + ".ff1.b"), Ref(), 0, 0, Ref(), 0, MVE_GEMM_GEGLU, "ff.geglu");
            rel(n3);
            Ref h4 = ws_stream((size_t)M * C * e);
            gemm(f, 4 * C, wt(b + ".ff2.w"), 4 * C, h4, C, M, C, 4 * C, wt(b + ".ff2.b"), Ref(), 0, 0, h, C, 0, "ff.out+residual");
            rel(f); rel(h); h = h4;
        }
        Ref out = ws_stream((size_t)M * C * e);
        gemm(h, C, wt(name + ".proj_out.w"), C, out, C, M, C, C, wt(name + ".proj_out.b"), Ref(), 0, 0, x, C, 0, "transformer.proj_out+residual");
        rel(h);
        return out;
    }

    // network-specific plan builders (builder_unet.h, builder_vae.h, builder_sr.h, builder_lpips.h)
    Ref vae_attention(const std::string& name, Ref x, int C, int H, int W);
    int build_lpips(int B_, int H, int W, int io_dtype);
    int build_sr(int B_, int H, int W, int io_dtype);
    int build_vae(int B_, int H, int W, int io_dtype);
    int build(int B_, int H, int W, int n_img, int has_res, int io_dtype, int res_nhwc);
};

}  // namespace
