// Fragment of the executor's single translation unit (csrc/unet.hip includes it; not a stand-alone header): slab layout of the packed parameters and the loader that packs one state-dict tensor into its place (every network mode).
#pragma once
#include "executor.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// parameter layout: reserve slab space for every packed tensor
// ---------------------------------------------------------------------------------------------------
struct SlabBuilder {
    Unet& u;
    size_t top = 0;
    void add(const std::string& name, size_t elems, bool f32, float q_fold = 0.f) {
        Param p;
        p.off = top; p.f32 = f32; p.bytes = elems * (f32 ? 4 : 2); p.q_fold = q_fold;
        top += (p.bytes + 255) & ~(size_t)255;
        u.params[name] = p;
    }
};

int g_fuse_shortcut = 1;     // engines created afterwards fold conv_shortcut into conv2 (mve_unet_tune; A/B measurements)

void layout_params(Unet& u) {
    const Config& c = u.cfg;
    SlabBuilder sb{u};
    const int T = c.temb_dim();
    std::vector<ResnetDesc> rs;
    std::vector<XfDesc> xs;
    enumerate(c, rs, xs);
    u.fuse_sc = g_fuse_shortcut != 0;
    for (int i = 0; i < c.n_levels; ++i) u.fuse_sc = u.fuse_sc && (c.ch[i] % 64 == 0);
    auto need = [&](const std::string& n) { u.expected.push_back(n); };
    if (c.lpips) {
        // every VGG conv twice: forward packing and the transposed / flipped packing that turns the same kernel into its dgrad
        for (int i = 0; i < 13; ++i) {
            const std::string n = vgg_name(i), e = "vgg." + std::to_string(i);
            const size_t ci = i == 0 ? 8 : VGG_CIN[i], co = VGG_COUT[i];
            sb.add(e + ".w", co * 9 * ci, false); need(n + ".weight");
            sb.add(e + ".wt", ci * 9 * co, false);
            sb.add(e + ".b", co, true); need(n + ".bias");
        }
        for (int k = 0; k < 5; ++k) { sb.add("lin." + std::to_string(k), VGG_COUT[VGG_BLK_FIRST[k + 1] - 1], true); need("lin" + std::to_string(k) + ".model.1.weight"); }
        sb.add("shift", 8, true); need("scaling_layer.shift");
        sb.add("scale", 8, true); need("scaling_layer.scale");
        sb.add("zeros", 512, true);          // ReLU = PReLU with zero slopes (the slab is zero-filled when it is allocated)
        u.slab_bytes = sb.top;
        return;
    }
    if (c.sr) {
        // body.0: conv in_ch -> F; body.(2k), k = 1..num_conv: conv F -> F; body.(2k+1): PReLU slopes; last: conv F -> out_ch * r * r
        const size_t F = c.ch[0];
        const int last = 2 * (c.layers_per_block + 1), opad = (c.out_ch * c.sr_scale * c.sr_scale + 7) & ~7;
        for (int k = 0; k <= c.layers_per_block + 1; ++k) {
            const std::string b = "body." + std::to_string(2 * k);
            const size_t rows = 2 * k == last ? (size_t)opad : F, cin = k == 0 ? 8 : F;
            sb.add(b + ".w", rows * 9 * cin, false); need(b + ".weight");
            sb.add(b + ".b", rows, true); need(b + ".bias");
            if (2 * k != last) { sb.add("body." + std::to_string(2 * k + 1) + ".a", F, true); need("body." + std::to_string(2 * k + 1) + ".weight"); }
        }
        u.slab_bytes = sb.top;
        return;
    }
    if (c.vae) {
        // names are the half's own (mve_unet_load_param strips `decoder.` / `encoder.`; (post_)quant_conv is `pq_conv`)
        const int n = c.n_levels, Cm = c.ch[n - 1], Cin0 = c.vae == 1 ? Cm : c.ch[0], Cout0 = c.vae == 1 ? c.ch[0] : Cm;
        sb.add("pq_conv.w", 8 * 8, false); need("pq_conv.weight");
        sb.add("pq_conv.b", 8, true); need("pq_conv.bias");
        sb.add("conv_in.w", (size_t)Cin0 * 9 * 8, false); need("conv_in.weight");
        sb.add("conv_in.b", Cin0, true); need("conv_in.bias");
        for (auto& r : rs) {
            sb.add(r.name + ".norm1.g", r.cin, true); need(r.name + ".norm1.weight");
            sb.add(r.name + ".norm1.b", r.cin, true); need(r.name + ".norm1.bias");
            sb.add(r.name + ".conv1.w", (size_t)r.cout * 9 * r.cin, false); need(r.name + ".conv1.weight");
            sb.add(r.name + ".conv1.b", r.cout, true); need(r.name + ".conv1.bias");
            sb.add(r.name + ".norm2.g", r.cout, true); need(r.name + ".norm2.weight");
            sb.add(r.name + ".norm2.b", r.cout, true); need(r.name + ".norm2.bias");
            const bool sc = r.cin != r.cout;
            sb.add(r.name + ".conv2.w", (size_t)r.cout * (9 * r.cout + (sc && u.fuse_sc ? r.cin : 0)), false); need(r.name + ".conv2.weight");
            sb.add(r.name + ".conv2.b", r.cout, true); need(r.name + ".conv2.bias");
            if (sc) {
                u.sc_cin[r.name] = r.cin;
                if (!u.fuse_sc) sb.add(r.name + ".sc.w", (size_t)r.cout * r.cin, false);
                need(r.name + ".conv_shortcut.weight");
                sb.add(r.name + ".sc.b", r.cout, true); need(r.name + ".conv_shortcut.bias");
            }
        }
        const std::string a = "mid_block.attentions.0";
        const size_t C = Cm;
        sb.add(a + ".group_norm.g", C, true); need(a + ".group_norm.weight");
        sb.add(a + ".group_norm.b", C, true); need(a + ".group_norm.bias");
        sb.add(a + ".qk.w", 2 * C * C, false); need(a + ".to_q.weight"); need(a + ".to_k.weight");
        sb.add(a + ".qk.b", 2 * C, true); need(a + ".to_q.bias"); need(a + ".to_k.bias");
        sb.add(a + ".v.w", C * C, false); need(a + ".to_v.weight");
        sb.add(a + ".v.b", C, true); need(a + ".to_v.bias");
        sb.add(a + ".o.w", C * C, false); need(a + ".to_out.0.weight");
        sb.add(a + ".o.b", C, true); need(a + ".to_out.0.bias");
        for (int i = 0; i + 1 < n; ++i) {
            const size_t Cs = c.vae == 1 ? c.ch[n - 1 - i] : c.ch[i];
            const std::string sn = (c.vae == 1 ? "up_blocks." + std::to_string(i) + ".upsamplers" : "down_blocks." + std::to_string(i) + ".downsamplers") + ".0.conv";
            sb.add(sn + ".w", Cs * 9 * Cs, false); need(sn + ".weight");
            if (c.vae == 1 && Cs % 64 == 0) sb.add(sn + ".w4", Cs * 16 * Cs, false);      // the upsampler's summed taps (mve_upsample_conv_phases)
            sb.add(sn + ".b", Cs, true); need(sn + ".bias");
        }
        sb.add("norm_out.g", Cout0, true); need("conv_norm_out.weight");
        sb.add("norm_out.b", Cout0, true); need("conv_norm_out.bias");
        sb.add("conv_out.w", (size_t)8 * 9 * Cout0, false); need("conv_out.weight");
        sb.add("conv_out.b", 8, true); need("conv_out.bias");
        u.slab_bytes = sb.top;
        return;
    }
    sb.add("conv_in.w", (size_t)c.ch[0] * 9 * 8, false); need("conv_in.weight");
    sb.add("conv_in.b", c.ch[0], true); need("conv_in.bias");
    sb.add("time.w1", (size_t)T * c.ch[0], false); need("time_embedding.linear_1.weight");
    sb.add("time.b1", T, true); need("time_embedding.linear_1.bias");
    sb.add("time.w2", (size_t)T * T, false); need("time_embedding.linear_2.weight");
    sb.add("time.b2", T, true); need("time_embedding.linear_2.bias");
    u.sum_temb = 0;
    for (auto& r : rs) { u.temb_off[r.name] = u.sum_temb; u.sum_temb += r.cout; }
    sb.add("temb_proj.w", (size_t)u.sum_temb * T, false);
    sb.add("temb_proj.b", u.sum_temb, true);
    for (auto& r : rs) {
        sb.add(r.name + ".norm1.g", r.cin, true); need(r.name + ".norm1.weight");
        sb.add(r.name + ".norm1.b", r.cin, true); need(r.name + ".norm1.bias");
        sb.add(r.name + ".conv1.w", (size_t)r.cout * 9 * r.cin, false); need(r.name + ".conv1.weight");
        sb.add(r.name + ".conv1.b", r.cout, true); need(r.name + ".conv1.bias");
        need(r.name + ".time_emb_proj.weight"); need(r.name + ".time_emb_proj.bias");
        sb.add(r.name + ".norm2.g", r.cout, true); need(r.name + ".norm2.weight");
        sb.add(r.name + ".norm2.b", r.cout, true); need(r.name + ".norm2.bias");
        const bool sc = r.cin != r.cout;
        sb.add(r.name + ".conv2.w", (size_t)r.cout * (9 * r.cout + (sc && u.fuse_sc ? r.cin : 0)), false); need(r.name + ".conv2.weight");
        sb.add(r.name + ".conv2.b", r.cout, true); need(r.name + ".conv2.bias");
        if (sc) {
            u.sc_cin[r.name] = r.cin;
            if (!u.fuse_sc) sb.add(r.name + ".sc.w", (size_t)r.cout * r.cin, false);
            need(r.name + ".conv_shortcut.weight");
            sb.add(r.name + ".sc.b", r.cout, true); need(r.name + ".conv_shortcut.bias");
        }
    }
    u.sum_kv = 0;
    for (auto& x : xs)
        for (int k = 0; k < x.layers; ++k) {
            u.kv_off[x.name + ".transformer_blocks." + std::to_string(k)] = u.sum_kv;
            u.sum_kv += 2 * x.c;
        }
    sb.add("ctx_kv.w", (size_t)u.sum_kv * c.ctx_dim, false);
    sb.add("ip_kv.w", (size_t)u.sum_kv * c.ctx_dim, false);      // IP-Adapter to_k_ip / to_v_ip, same column layout (optional weights)
    u.n_xf_layers = (int)u.kv_off.size();
    for (auto& x : xs) {
        const size_t C = x.c;
        sb.add(x.name + ".norm.g", C, true); need(x.name + ".norm.weight");
        sb.add(x.name + ".norm.b", C, true); need(x.name + ".norm.bias");
        sb.add(x.name + ".proj_in.w", C * C, false); need(x.name + ".proj_in.weight");
        sb.add(x.name + ".proj_in.b", C, true); need(x.name + ".proj_in.bias");
        sb.add(x.name + ".proj_out.w", C * C, false); need(x.name + ".proj_out.weight");
        sb.add(x.name + ".proj_out.b", C, true); need(x.name + ".proj_out.bias");
        for (int k = 0; k < x.layers; ++k) {
            const std::string b = x.name + ".transformer_blocks." + std::to_string(k);
            for (const char* nn : {"norm1", "norm2", "norm3"}) {
                sb.add(b + "." + nn + ".g", C, true); need(b + "." + nn + ".weight");
                sb.add(b + "." + nn + ".b", C, true); need(b + "." + nn + ".bias");
            }
            // to_q rows carry softmax_scale * log2(e) = head_dim^-1/2 * log2(e): recorded on the slot, applied by load_param, required by Builder::attn
            const float qf = 1.4426950408889634f / sqrtf((float)(C / x.heads));
            sb.add(b + ".qkv.w", 3 * C * C, false, qf);
            need(b + ".attn1.to_q.weight"); need(b + ".attn1.to_k.weight"); need(b + ".attn1.to_v.weight");
            sb.add(b + ".o1.w", C * C, false); need(b + ".attn1.to_out.0.weight");
            sb.add(b + ".o1.b", C, true); need(b + ".attn1.to_out.0.bias");
            sb.add(b + ".q2.w", C * C, false, qf); need(b + ".attn2.to_q.weight");
            need(b + ".attn2.to_k.weight"); need(b + ".attn2.to_v.weight");
            sb.add(b + ".o2.w", C * C, false); need(b + ".attn2.to_out.0.weight");
            sb.add(b + ".o2.b", C, true); need(b + ".attn2.to_out.0.bias");
            sb.add(b + ".ff1.w", 8 * C * C, false); need(b + ".ff.net.0.proj.weight");
            sb.add(b + ".ff1.b", 8 * C, true); need(b + ".ff.net.0.proj.bias");
            sb.add(b + ".ff2.w", 4 * C * C, false); need(b + ".ff.net.2.weight");
            sb.add(b + ".ff2.b", C, true); need(b + ".ff.net.2.bias");
        }
    }
    for (int i = 0; i + 1 < c.n_levels; ++i) {
        const size_t C = c.ch[i];
        const std::string d = "down_blocks." + std::to_string(i) + ".downsamplers.0.conv";
        sb.add(d + ".w", C * 9 * C, false); need(d + ".weight");
        sb.add(d + ".b", C, true); need(d + ".bias");
        if (c.controlnet) continue;
        const size_t Cu = c.ch[c.n_levels - 1 - i];
        const std::string up = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
        sb.add(up + ".w", Cu * 9 * Cu, false); need(up + ".weight");
        if (Cu % 64 == 0) sb.add(up + ".w4", Cu * 16 * Cu, false);      // the summed taps of the four 2 x 2 phase convs (mve_upsample_conv_phases)
        sb.add(up + ".b", Cu, true); need(up + ".bias");
    }
    if (!c.controlnet) {
        sb.add("norm_out.g", c.ch[0], true); need("conv_norm_out.weight");
        sb.add("norm_out.b", c.ch[0], true); need("conv_norm_out.bias");
        sb.add("conv_out.w", (size_t)8 * 9 * c.ch[0], false); need("conv_out.weight");
        sb.add("conv_out.b", 8, true); need("conv_out.bias");
    } else {
        // controlnet_cond_embedding: conv_in (cond_ch -> 16), blocks (16->16, 16->32 s2, 32->32, 32->96 s2, 96->96, 96->256 s2),
        // conv_out (256 -> ch[0]); controlnet_down_blocks.k / controlnet_mid_block: 1x1 "zero" convolutions
        const std::string e = "controlnet_cond_embedding.";
        sb.add(e + "conv_in.w", (size_t)CN_EMB[0] * 9 * 8, false); need(e + "conv_in.weight");
        sb.add(e + "conv_in.b", CN_EMB[0], true); need(e + "conv_in.bias");
        for (int k = 0; k < 6; ++k) {
            const int ci = CN_EMB[k / 2], co = CN_EMB[(k + 1) / 2];
            const std::string b = e + "blocks." + std::to_string(k);
            sb.add(b + ".w", (size_t)co * 9 * ci, false); need(b + ".weight");
            sb.add(b + ".b", co, true); need(b + ".bias");
        }
        sb.add(e + "conv_out.w", (size_t)c.ch[0] * 9 * CN_EMB[3], false); need(e + "conv_out.weight");
        sb.add(e + "conv_out.b", c.ch[0], true); need(e + "conv_out.bias");
        int k = 0;
        auto zero_conv = [&](int C) {
            const std::string z = "controlnet_down_blocks." + std::to_string(k++);
            sb.add(z + ".w", (size_t)C * C, false); need(z + ".weight");
            sb.add(z + ".b", C, true); need(z + ".bias");
        };
        zero_conv(c.ch[0]);
        for (int i = 0; i < c.n_levels; ++i) {
            for (int j = 0; j < c.layers_per_block; ++j) zero_conv(c.ch[i]);
            if (i + 1 < c.n_levels) zero_conv(c.ch[i]);
        }
        const size_t Cm = c.ch[c.n_levels - 1];
        sb.add("controlnet_mid_block.w", Cm * Cm, false); need("controlnet_mid_block.weight");
        sb.add("controlnet_mid_block.b", Cm, true); need("controlnet_mid_block.bias");
    }
    u.slab_bytes = sb.top;
}

bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
std::string strip(const std::string& s, const std::string& suf) { return s.substr(0, s.size() - suf.size()); }

// ---------------------------------------------------------------------------------------------------
// load one diffusers tensor into its packed place
// ---------------------------------------------------------------------------------------------------
int load_param(Unet& u, const std::string& name, const void* src, int src_dtype, int ndim, const long long* shape,
               hipStream_t s) {
    const Config& c = u.cfg;
    auto P = [&](const std::string& n) -> Param* {
        auto it = u.params.find(n);
        return it == u.params.end() ? nullptr : &it->second;
    };
    auto dstp = [&](Param* p, size_t elem_off) { return (void*)(u.slab + p->off + elem_off * (p->f32 ? 4 : 2)); };
    auto numel = [&]() { long long n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; };
    PackDims d;
    auto vec = [&](Param* p, size_t off, long long n, long long dst_stride) -> int {   // 1-D copy to f32/16-bit, strided dst
        MVE_CHECK(p, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
        MVE_CHECK(numel() == n, MVE_ERR_ARG, "load_param(%s): expected %lld elements, got %lld", name.c_str(), n, numel());
        d = PackDims{{1, 1, 1, n}, {0, 0, 0, 1}, {0, 0, 0, dst_stride}, n};
        return pack(src_dtype, p->f32 ? MVE_F32 : c.dtype, src, dstp(p, off), d, s);
    };
    auto mat = [&](Param* p, size_t elem_off, long long N, long long K, long long dst_row_stride, float mul = 1.0f) -> int {   // [N][K] rows
        MVE_CHECK(p, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
        MVE_CHECK(numel() == N * K, MVE_ERR_ARG, "load_param(%s): expected %lldx%lld, got %lld elements", name.c_str(), N, K, numel());
        d = PackDims{{1, 1, N, K}, {0, 0, K, 1}, {0, 0, dst_row_stride, 1}, K};
        d.mul = mul;
        return pack(src_dtype, c.dtype, src, dstp(p, elem_off), d, s);
    };
    // to_q rows carry softmax_scale * log2(e) = head_dim^-1/2 * log2(e): the attention kernel then works in log2 units without a multiply per
    // logit (mve_attention_prescaled).  One rounding either way: the factor is applied in fp32 before the weight is rounded to 16 bits.
    // The factor lives on the destination slot (layout_params): no second walk over the configuration per tensor.
    auto q_fold = [&](const std::string& slot) -> float {
        auto it = u.params.find(slot);
        return it == u.params.end() ? 0.f : it->second.q_fold;
    };
    long long conv_row = 0;    // destination row length of the conv packer when the row also holds a fused shortcut (0: 9 * I)
    auto conv = [&](Param* p, long long O, long long I, long long Opad, long long Ipad) -> int {   // OIHW -> [O][3][3][Ipad]
        MVE_CHECK(p, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
        MVE_CHECK(ndim == 4 && shape[0] == O && shape[1] == I && shape[2] == 3 && shape[3] == 3, MVE_ERR_ARG,
                  "load_param(%s): expected [%lld,%lld,3,3]", name.c_str(), O, I);
        (void)Opad;
        if (I % 64 == 0 && Ipad == I)   // channel-slab-major K order [O][I/64][9][64] (MVE_CONV_W_CHUNK64)
            d = PackDims{{O, I / 64, 9, 64}, {I * 9, 64 * 9, 1, 9}, {conv_row ? conv_row : 9 * I, 9 * 64, 64, 1}, 64};
        else
            d = PackDims{{O, 3, 3, Ipad}, {I * 9, 3, 1, 9}, {9 * Ipad, 3 * Ipad, Ipad, 1}, I};
        return pack(src_dtype, c.dtype, src, dstp(p, 0), d, s);
    };
    int rc = MVE_ERR_ARG;
    const int T = c.temb_dim();
    const int Cm_ = c.ch[c.n_levels - 1];
    const int vin = c.vae == 1 ? Cm_ : c.ch[0], vout = c.vae == 1 ? c.ch[0] : Cm_;     // widths after conv_in / before conv_out
    const std::string va = "mid_block.attentions.0";
    if (c.lpips) {
        int li = -1;
        for (int i = 0; i < 13; ++i) if (name.compare(0, vgg_name(i).size() + 1, vgg_name(i) + ".") == 0) li = i;
        if (li >= 0 && ends_with(name, ".weight")) {
            const long long co = VGG_COUT[li], ci = VGG_CIN[li];
            const std::string e = "vgg." + std::to_string(li);
            MVE_CHECK(ndim == 4 && shape[0] == co && shape[1] == ci && shape[2] == 3 && shape[3] == 3, MVE_ERR_ARG, "load_param(%s): expected [%lld,%lld,3,3]", name.c_str(), co, ci);
            if (li == 0) {
                MVE_HIP(hipMemsetAsync(dstp(P(e + ".w"), 0), 0, P(e + ".w")->bytes, s));
                rc = conv(P(e + ".w"), co, ci, co, 8);
                MVE_HIP(hipMemsetAsync(dstp(P(e + ".wt"), 0), 0, P(e + ".wt")->bytes, s));
            } else rc = conv(P(e + ".w"), co, ci, co, ci);
            if (rc == MVE_OK) {
                // dgrad weight as a conv weight: W'[o' = ci][i' = co][ky][kx] = W[co][ci][2-ky][2-kx]; co is always a multiple of 64, so
                // the slab-major layout [O'][I'/64][9][64]; source offset of (o', slab, tap, c) = (slab*64 + c)*ci*9 + o'*9 + (8 - tap)
                const long long rows = li == 0 ? 8 : ci;         // conv1_1: 3 real rows, padded to 8 (the rest stay zero)
                (void)rows;
                PackDims dd{{ci, co / 64, 9, 64}, {9, 64 * ci * 9, -1, ci * 9}, {9 * co, 9 * 64, 64, 1}, 64};
                const size_t esz_src = src_dtype == MVE_F32 ? 4 : 2;
                rc = pack(src_dtype, c.dtype, (const unsigned char*)src + 8 * esz_src, dstp(P(e + ".wt"), 0), dd, s);
            }
        } else if (li >= 0 && ends_with(name, ".bias")) rc = vec(P("vgg." + std::to_string(li) + ".b"), 0, VGG_COUT[li], 1);
        else if (name == "scaling_layer.shift" || name == "scaling_layer.scale") rc = vec(P(name.substr(14)), 0, 3, 1);
        else if (name.compare(0, 3, "lin") == 0 && ends_with(name, ".model.1.weight")) {
            const int k = name[3] - '0';
            MVE_CHECK(k >= 0 && k < 5, MVE_ERR_ARG, "load_param: no layer %s", name.c_str());
            rc = vec(P("lin." + std::to_string(k)), 0, VGG_COUT[VGG_BLK_FIRST[k + 1] - 1], 1);
        } else {
            mve_set_error("load_param: %s is not a parameter of LPIPS(net='vgg')", name.c_str());
            return MVE_ERR_ARG;
        }
    } else if (c.sr) {
        MVE_CHECK(name.compare(0, 5, "body.") == 0, MVE_ERR_ARG, "load_param: %s is not a parameter of SRVGGNetCompact", name.c_str());
        const int idx = atoi(name.c_str() + 5), last = 2 * (c.layers_per_block + 1);
        const long long F = c.ch[0], nout = (long long)c.out_ch * c.sr_scale * c.sr_scale;
        const std::string b = "body." + std::to_string(idx);
        MVE_CHECK(idx >= 0 && idx <= last && name.size() > b.size(), MVE_ERR_ARG, "load_param: no layer %s", name.c_str());
        const std::string leaf = name.substr(b.size());
        if (idx % 2 == 1 && leaf == ".weight") rc = vec(P(b + ".a"), 0, F, 1);                      // PReLU slopes
        else if (idx % 2 == 0 && leaf == ".weight") {
            Param* pw = P(b + ".w");
            MVE_CHECK(pw, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
            MVE_HIP(hipMemsetAsync(dstp(pw, 0), 0, pw->bytes, s));
            rc = idx == 0 ? conv(pw, F, c.in_ch, F, 8) : conv(pw, idx == last ? nout : F, F, 0, F);
        } else if (idx % 2 == 0 && leaf == ".bias") {
            Param* pb = P(b + ".b");
            MVE_CHECK(pb, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
            MVE_HIP(hipMemsetAsync(dstp(pb, 0), 0, pb->bytes, s));
            rc = vec(pb, 0, idx == last ? nout : F, 1);
        } else {
            mve_set_error("load_param: %s is not a parameter of SRVGGNetCompact", name.c_str());
            return MVE_ERR_ARG;
        }
    } else if (c.vae && name == "conv_in.weight") {
        MVE_HIP(hipMemsetAsync(dstp(P("conv_in.w"), 0), 0, P("conv_in.w")->bytes, s));
        rc = conv(P("conv_in.w"), vin, c.in_ch, vin, 8);
    } else if (c.vae && name == "conv_in.bias") rc = vec(P("conv_in.b"), 0, vin, 1);
    else if (c.vae && name == "conv_norm_out.weight") rc = vec(P("norm_out.g"), 0, vout, 1);
    else if (c.vae && name == "conv_norm_out.bias") rc = vec(P("norm_out.b"), 0, vout, 1);
    else if (c.vae && name == "conv_out.weight") {
        MVE_HIP(hipMemsetAsync(dstp(P("conv_out.w"), 0), 0, P("conv_out.w")->bytes, s));
        rc = conv(P("conv_out.w"), c.out_ch, vout, 8, vout);
    } else if (c.vae && name == "pq_conv.weight") {      // (post_)quant_conv: 1x1 over <= 8 channels, zero-padded to 8 x 8
        const int nq = c.vae == 1 ? c.in_ch : c.out_ch;
        MVE_HIP(hipMemsetAsync(dstp(P("pq_conv.w"), 0), 0, P("pq_conv.w")->bytes, s));
        rc = mat(P("pq_conv.w"), 0, nq, nq, 8);
    } else if (c.vae && name == "pq_conv.bias") {
        MVE_HIP(hipMemsetAsync(dstp(P("pq_conv.b"), 0), 0, P("pq_conv.b")->bytes, s));
        rc = vec(P("pq_conv.b"), 0, c.vae == 1 ? c.in_ch : c.out_ch, 1);
    } else if (c.vae && (name == va + ".to_q.weight" || name == va + ".to_k.weight"))
        rc = mat(P(va + ".qk.w"), name[va.size() + 4] == 'q' ? 0 : (size_t)Cm_ * Cm_, Cm_, Cm_, Cm_);
    else if (c.vae && (name == va + ".to_q.bias" || name == va + ".to_k.bias"))
        rc = vec(P(va + ".qk.b"), name[va.size() + 4] == 'q' ? 0 : Cm_, Cm_, 1);
    else if (c.vae && name == va + ".to_v.weight") rc = mat(P(va + ".v.w"), 0, Cm_, Cm_, Cm_);
    else if (c.vae && name == va + ".to_v.bias") rc = vec(P(va + ".v.b"), 0, Cm_, 1);
    else if (c.vae && name == va + ".to_out.0.weight") rc = mat(P(va + ".o.w"), 0, Cm_, Cm_, Cm_);
    else if (c.vae && name == va + ".to_out.0.bias") rc = vec(P(va + ".o.b"), 0, Cm_, 1);
    else if (name == "conv_in.weight") {
        MVE_HIP(hipMemsetAsync(dstp(P("conv_in.w"), 0), 0, P("conv_in.w")->bytes, s));
        rc = conv(P("conv_in.w"), c.ch[0], c.in_ch, c.ch[0], 8);
    } else if (name == "conv_in.bias") rc = vec(P("conv_in.b"), 0, c.ch[0], 1);
    else if (name == "time_embedding.linear_1.weight") rc = mat(P("time.w1"), 0, T, c.ch[0], c.ch[0]);
    else if (name == "time_embedding.linear_1.bias") rc = vec(P("time.b1"), 0, T, 1);
    else if (name == "time_embedding.linear_2.weight") rc = mat(P("time.w2"), 0, T, T, T);
    else if (name == "time_embedding.linear_2.bias") rc = vec(P("time.b2"), 0, T, 1);
    else if (name == "conv_norm_out.weight") rc = vec(P("norm_out.g"), 0, c.ch[0], 1);
    else if (name == "conv_norm_out.bias") rc = vec(P("norm_out.b"), 0, c.ch[0], 1);
    else if (name == "conv_out.weight") {
        MVE_HIP(hipMemsetAsync(dstp(P("conv_out.w"), 0), 0, P("conv_out.w")->bytes, s));
        rc = conv(P("conv_out.w"), c.out_ch, c.ch[0], 8, c.ch[0]);
    } else if (name == "conv_out.bias") {
        MVE_HIP(hipMemsetAsync(dstp(P("conv_out.b"), 0), 0, P("conv_out.b")->bytes, s));
        rc = vec(P("conv_out.b"), 0, c.out_ch, 1);
    } else if (ends_with(name, ".time_emb_proj.weight")) {
        const std::string r = strip(name, ".time_emb_proj.weight");
        MVE_CHECK(u.temb_off.count(r), MVE_ERR_ARG, "load_param: unknown resnet %s", r.c_str());
        const long long cout = shape[0];
        rc = mat(P("temb_proj.w"), (size_t)u.temb_off[r] * T, cout, T, T);
    } else if (ends_with(name, ".time_emb_proj.bias")) {
        const std::string r = strip(name, ".time_emb_proj.bias");
        MVE_CHECK(u.temb_off.count(r), MVE_ERR_ARG, "load_param: unknown resnet %s", r.c_str());
        rc = vec(P("temb_proj.b"), u.temb_off[r], shape[0], 1);
    } else if (name == "controlnet_cond_embedding.conv_in.weight") {
        MVE_HIP(hipMemsetAsync(dstp(P("controlnet_cond_embedding.conv_in.w"), 0), 0, P("controlnet_cond_embedding.conv_in.w")->bytes, s));
        rc = conv(P("controlnet_cond_embedding.conv_in.w"), CN_EMB[0], c.cond_ch, CN_EMB[0], 8);
    } else if (name.compare(0, 26, "controlnet_cond_embedding.") == 0 && ends_with(name, ".weight")) {
        MVE_CHECK(ndim == 4, MVE_ERR_ARG, "load_param(%s): expected a 4-D conv weight", name.c_str());
        rc = conv(P(strip(name, ".weight") + ".w"), shape[0], shape[1], shape[0], shape[1]);
    } else if (name.compare(0, 26, "controlnet_cond_embedding.") == 0 && ends_with(name, ".bias")) {
        rc = vec(P(strip(name, ".bias") + ".b"), 0, shape[0], 1);
    } else if ((name.compare(0, 23, "controlnet_down_blocks.") == 0 || name.compare(0, 21, "controlnet_mid_block.") == 0) && ends_with(name, ".weight")) {
        Param* p = P(strip(name, ".weight") + ".w");
        MVE_CHECK(p && ndim >= 2, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
        rc = mat(p, 0, shape[0], shape[1], shape[1]);
    } else if ((name.compare(0, 23, "controlnet_down_blocks.") == 0 || name.compare(0, 21, "controlnet_mid_block.") == 0) && ends_with(name, ".bias")) {
        rc = vec(P(strip(name, ".bias") + ".b"), 0, shape[0], 1);
    } else if (ends_with(name, ".conv_shortcut.weight") && u.fuse_sc) {
        const std::string r = strip(name, ".conv_shortcut.weight");
        MVE_CHECK(u.sc_cin.count(r) && ndim >= 2 && shape[1] == u.sc_cin[r], MVE_ERR_ARG, "load_param: unexpected shortcut %s", name.c_str());
        const long long cout = shape[0], cin = shape[1];
        rc = mat(P(r + ".conv2.w"), (size_t)9 * cout, cout, cin, 9 * cout + cin);      // tail columns of every conv2 row
    } else if (ends_with(name, ".conv_shortcut.weight")) {
        Param* p = P(strip(name, ".conv_shortcut.weight") + ".sc.w");
        MVE_CHECK(p && ndim >= 2, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
        rc = mat(p, 0, shape[0], shape[1], shape[1]);
    } else if (ends_with(name, ".conv_shortcut.bias")) rc = vec(P(strip(name, ".conv_shortcut.bias") + ".sc.b"), 0, shape[0], 1);
    else if (ends_with(name, ".conv1.weight") || ends_with(name, ".conv2.weight") || ends_with(name, ".conv.weight")) {
        const std::string base = strip(name, ".weight");
        MVE_CHECK(ndim == 4, MVE_ERR_ARG, "load_param(%s): expected a 4-D conv weight", name.c_str());
        if (u.fuse_sc && ends_with(name, ".conv2.weight")) {
            const std::string r = strip(name, ".conv2.weight");
            if (u.sc_cin.count(r)) conv_row = 9 * shape[1] + u.sc_cin[r];
        }
        rc = conv(P(base + ".w"), shape[0], shape[1], shape[0], shape[1]);
        if (rc == MVE_OK && P(base + ".w4"))      // an upsampler: also the summed taps of its four 2 x 2 phase convs
            rc = mve_pack_upsample_phase_weights(src_dtype, c.dtype, src, (int)shape[0], (int)shape[1], dstp(P(base + ".w4"), 0), s);
    } else if (ends_with(name, ".conv1.bias") || ends_with(name, ".conv2.bias") || ends_with(name, ".conv.bias"))
        rc = vec(P(strip(name, ".bias") + ".b"), 0, shape[0], 1);
    else if (ends_with(name, ".proj_in.weight") || ends_with(name, ".proj_out.weight")) {
        Param* p = P(strip(name, ".weight") + ".w");
        MVE_CHECK(p && ndim >= 2, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
        rc = mat(p, 0, shape[0], shape[1], shape[1]);   // [C,C] or [C,C,1,1]
    } else if (ends_with(name, ".proj_in.bias") || ends_with(name, ".proj_out.bias"))
        rc = vec(P(strip(name, ".bias") + ".b"), 0, shape[0], 1);
    else if (ends_with(name, ".attn1.to_q.weight") || ends_with(name, ".attn1.to_k.weight") || ends_with(name, ".attn1.to_v.weight")) {
        const int which = name[name.size() - 8] == 'q' ? 0 : (name[name.size() - 8] == 'k' ? 1 : 2);
        const std::string b = name.substr(0, name.size() - std::string(".attn1.to_q.weight").size());
        const long long C = shape[0];
        const float fold = which == 0 ? q_fold(b + ".qkv.w") : 1.0f;
        MVE_CHECK(fold > 0.f, MVE_ERR_ARG, "load_param: %s is not in a known transformer block", name.c_str());
        rc = mat(P(b + ".qkv.w"), (size_t)which * C * C, C, C, C, fold);
    } else if (ends_with(name, ".attn2.to_k.weight") || ends_with(name, ".attn2.to_v.weight")) {
        const int which = name[name.size() - 8] == 'k' ? 0 : 1;
        const std::string b = name.substr(0, name.size() - std::string(".attn2.to_k.weight").size());
        MVE_CHECK(u.kv_off.count(b), MVE_ERR_ARG, "load_param: unknown attention block %s", b.c_str());
        const long long C = shape[0];
        rc = mat(P("ctx_kv.w"), ((size_t)u.kv_off[b] + (size_t)which * C) * c.ctx_dim, C, c.ctx_dim, c.ctx_dim);
    } else if (ends_with(name, ".attn2.processor.to_k_ip.weight") || ends_with(name, ".attn2.processor.to_v_ip.weight")) {
        const int which = name[name.size() - 11] == 'k' ? 0 : 1;
        const std::string b = name.substr(0, name.size() - std::string(".attn2.processor.to_k_ip.weight").size());
        MVE_CHECK(u.kv_off.count(b), MVE_ERR_ARG, "load_param: unknown attention block %s", b.c_str());
        const long long C = shape[0];
        rc = mat(P("ip_kv.w"), ((size_t)u.kv_off[b] + (size_t)which * C) * c.ctx_dim, C, c.ctx_dim, c.ctx_dim);
        if (rc == MVE_OK && !u.loaded.count(name)) ++u.n_ip_loaded;
    } else if (ends_with(name, ".attn2.to_q.weight")) {
        const float fold = q_fold(strip(name, ".attn2.to_q.weight") + ".q2.w");
        MVE_CHECK(fold > 0.f, MVE_ERR_ARG, "load_param: %s is not in a known transformer block", name.c_str());
        rc = mat(P(strip(name, ".attn2.to_q.weight") + ".q2.w"), 0, shape[0], shape[1], shape[1], fold);
    }
    else if (ends_with(name, ".attn1.to_out.0.weight")) rc = mat(P(strip(name, ".attn1.to_out.0.weight") + ".o1.w"), 0, shape[0], shape[1], shape[1]);
    else if (ends_with(name, ".attn1.to_out.0.bias")) rc = vec(P(strip(name, ".attn1.to_out.0.bias") + ".o1.b"), 0, shape[0], 1);
    else if (ends_with(name, ".attn2.to_out.0.weight")) rc = mat(P(strip(name, ".attn2.to_out.0.weight") + ".o2.w"), 0, shape[0], shape[1], shape[1]);
    else if (ends_with(name, ".attn2.to_out.0.bias")) rc = vec(P(strip(name, ".attn2.to_out.0.bias") + ".o2.b"), 0, shape[0], 1);
    else if (ends_with(name, ".ff.net.0.proj.weight")) {
        // rows [0,4C) = value, [4C,8C) = gate  ->  interleaved (value_i, gate_i)
        Param* p = P(strip(name, ".ff.net.0.proj.weight") + ".ff1.w");
        MVE_CHECK(p && ndim == 2 && shape[0] % 2 == 0, MVE_ERR_ARG, "load_param: bad %s", name.c_str());
        const long long half = shape[0] / 2, K = shape[1];
        d = PackDims{{1, 2, half, K}, {0, half * K, K, 1}, {0, K, 2 * K, 1}, K};
        rc = pack(src_dtype, c.dtype, src, dstp(p, 0), d, s);
    } else if (ends_with(name, ".ff.net.0.proj.bias")) {
        Param* p = P(strip(name, ".ff.net.0.proj.bias") + ".ff1.b");
        MVE_CHECK(p && shape[0] % 2 == 0, MVE_ERR_ARG, "load_param: bad %s", name.c_str());
        const long long half = shape[0] / 2;
        d = PackDims{{1, 1, 2, half}, {0, 0, half, 1}, {0, 0, 1, 2}, half};
        rc = pack(src_dtype, MVE_F32, src, dstp(p, 0), d, s);
    } else if (ends_with(name, ".ff.net.2.weight")) rc = mat(P(strip(name, ".ff.net.2.weight") + ".ff2.w"), 0, shape[0], shape[1], shape[1]);
    else if (ends_with(name, ".ff.net.2.bias")) rc = vec(P(strip(name, ".ff.net.2.bias") + ".ff2.b"), 0, shape[0], 1);
    else if (ends_with(name, ".weight") && P(strip(name, ".weight") + ".g")) rc = vec(P(strip(name, ".weight") + ".g"), 0, shape[0], 1);   // norms
    else if (ends_with(name, ".bias") && P(strip(name, ".bias") + ".b")) rc = vec(P(strip(name, ".bias") + ".b"), 0, shape[0], 1);
    else {
        mve_set_error("load_param: %s is not a parameter of this UNet configuration", name.c_str());
        return MVE_ERR_ARG;
    }
    if (rc == MVE_OK) u.loaded[name] = true;
    return rc;
}

}  // namespace
