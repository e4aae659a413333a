// UNet2DCondition executor: a native (C++) runtime behind the reference's `self.unet(...)` seam
// (lib/pipelines/adapter3d_mixin.py:117-125) and its two halves `unet_enc` / `unet_dec`
// (lib/models/architecture/diffusers.py:57-99, :102-164).
//
// Design (MI355X first, not a translation of diffusers' module tree):
//   * activations live in one caller-provided workspace, NHWC ([B*H*W, C] row-major, fp16/bf16), laid out by
//     a plan-time allocator with explicit lifetimes -- tokens for attention and pixels for convolution are
//     the same memory, so the reference's permute/reshape/contiguous traffic does not exist;
//   * the whole forward is a static op list ("plan") built once per (batch, H, W, n_cross_img): every op is
//     one launch of a kernel from gemm.hip / attention.hip / norm.hip / elementwise.hip on one stream;
//   * weights are engine-owned and packed at load time from diffusers' state-dict layout by a device
//     kernel (OIHW -> OHWI, q/k/v fused into one [3C,C] matrix, GEGLU value/gate rows interleaved so the
//     gate is applied in the GEMM epilogue, conv_in/conv_out channels padded 4 -> 8);
//   * work that only depends on (t, text) is hoisted and batched: ONE GEMM produces the time-embedding
//     projections of all ResnetBlocks, ONE GEMM produces K and V of all cross-attention layers;
//   * skip-concats are never materialised except as the GroupNorm output the next conv reads anyway;
//   * per-op HIP-event profiling and analytic FLOP accounting are built in (bench.py's roofline).
#include "common.h"

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

extern "C" {
int mve_gemm(int, const void*, int, const void*, int, void*, int, int, int, int, const float*, const float*, int, int,
             const void*, int, int, float, void*, size_t, int, void*);
int mve_conv3x3(int, const void*, int, const void*, int, int, int, int, int, int, const void*, int, void*, int,
                const float*, const float*, int, const void*, int, int, float, void*, size_t, void*);
size_t mve_gemm_workspace_bytes(int, int, int, int);
int mve_attention(int, const void*, int, const void*, int, const void*, int, const void*, int, const void*, int, void*, int,
                  int, int, int, int, int, int, float, void*);
size_t mve_groupnorm_workspace_bytes(int, int, int, int);
int mve_groupnorm_silu(int, const void*, int, const void*, int, int, int, int, float, const float*, const float*, int, void*,
                       void*, void*);
int mve_layernorm(int, const void*, int, void*, int, int, int, const float*, const float*, float, void*);
int mve_nchw_to_nhwc(int, int, const void*, int, int, int, int, int, void*, void*);
int mve_nhwc_to_nchw(int, int, const void*, int, int, int, int, int, void*, void*);
int mve_timestep_embedding(int, const float*, int, int, void*, void*);
int mve_silu(int, const void*, void*, size_t, void*);
int mve_axpy(int, const void*, const void*, float, void*, size_t, void*);
int mve_softmax_rows(int, const float*, size_t, int, int, void*, size_t, void*);
int mve_prelu(int, const void*, const float*, int, void*, size_t, void*);
int mve_pixel_shuffle_add(int, const float*, int, const void*, int, int, int, int, int, void*, void*);
int mve_lpips_scale(int, int, const void*, const void*, int, int, int, const float*, const float*, int, void*, void*);
int mve_lpips_input_grad(int, int, const void*, int, int, int, const float*, int, void*, void*);
int mve_maxpool2x2(int, const void*, int, int, int, int, void*, void*);
int mve_maxpool2x2_backward(int, const void*, const void*, int, int, int, int, void*, void*);
int mve_relu_backward(int, void*, const void*, size_t, void*);
size_t mve_lpips_layer_scratch_bytes(int, int);
int mve_lpips_layer(int, const void*, const float*, int, int, int, int, float*, void*, void*);
int mve_lpips_layer_backward(int, const void*, const float*, const float*, int, int, int, void*, void*);
}

namespace {

constexpr int MAX_LEVELS = 8;

// ---------------------------------------------------------------------------------------------------
// weight packing kernel: dst[d0*t0 + d1*t1 + d2*t2 + d3*t3] = (d3 < valid3) ? src[d0*s0 + d1*s1 + d2*s2 + d3*s3] : 0
// ---------------------------------------------------------------------------------------------------
struct PackDims { long long D[4], s[4], t[4]; long long valid3; };

template <class Src, class Dst>
__global__ void k_pack(const Src* __restrict__ src, Dst* __restrict__ dst, PackDims p) {
    const long long n = p.D[0] * p.D[1] * p.D[2] * p.D[3];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const long long d3 = r % p.D[3]; r /= p.D[3];
        const long long d2 = r % p.D[2]; r /= p.D[2];
        const long long d1 = r % p.D[1]; r /= p.D[1];
        const long long d0 = r;
        float v = 0.f;
        if (d3 < p.valid3) v = (float)src[d0 * p.s[0] + d1 * p.s[1] + d2 * p.s[2] + d3 * p.s[3]];
        dst[d0 * p.t[0] + d1 * p.t[1] + d2 * p.t[2] + d3 * p.t[3]] = (Dst)v;
    }
}

template <class Src>
int pack_to(int dst_dtype, const void* src, void* dst, const PackDims& p, hipStream_t s) {
    const long long n = p.D[0] * p.D[1] * p.D[2] * p.D[3];
    if (n == 0) return MVE_OK;
    const unsigned grid = (unsigned)((n + 255) / 256 > 65535 ? 65535 : (n + 255) / 256);
    if (dst_dtype == MVE_F32) k_pack<Src, float><<<grid, 256, 0, s>>>((const Src*)src, (float*)dst, p);
    else if (dst_dtype == MVE_F16) k_pack<Src, f16><<<grid, 256, 0, s>>>((const Src*)src, (f16*)dst, p);
    else if (dst_dtype == MVE_BF16) k_pack<Src, bf16><<<grid, 256, 0, s>>>((const Src*)src, (bf16*)dst, p);
    else { mve_set_error("pack: bad dst dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int pack(int src_dtype, int dst_dtype, const void* src, void* dst, const PackDims& p, hipStream_t s) {
    if (src_dtype == MVE_F32) return pack_to<float>(dst_dtype, src, dst, p, s);
    if (src_dtype == MVE_F16) return pack_to<f16>(dst_dtype, src, dst, p, s);
    if (src_dtype == MVE_BF16) return pack_to<bf16>(dst_dtype, src, dst, p, s);
    mve_set_error("pack: bad src dtype %d", src_dtype);
    return MVE_ERR_ARG;
}

// mean over groups of n consecutive images: x [B, R] -> y [B/n, R]  (joint_attn.py:24 encoder_hidden_states_.mean(dim=1))
template <class Tag>
__global__ void k_group_mean(const typename Tag::T* __restrict__ x, typename Tag::T* __restrict__ y, long long R, int n,
                             long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long g = i / R, r = i - g * R;
    float a = 0.f;
    for (int k = 0; k < n; ++k) a += Tag::to_f32(x[(g * n + k) * R + r]);
    y[i] = Tag::from_f32(a / (float)n);
}

// ---------------------------------------------------------------------------------------------------
// configuration / parameter table
// ---------------------------------------------------------------------------------------------------
constexpr int CN_EMB[4] = {16, 32, 96, 256};      // diffusers ControlNetModel conditioning_embedding_out_channels (default)

struct Config {
    int controlnet = 0, cond_ch = 3;   // ControlNetModel: encoder + mid of the UNet, conditioning embedding, zero convolutions
    int lpips = 0, lpips_normalize = 1; // LPIPS(net='vgg') forward + backward w.r.t. the prediction (lib/models/losses/lpips_loss.py)
    int sr = 0, sr_scale = 4;          // SRVGGNetCompact (lib/models/decoders/image_space_ss.py): ch[0] = num_feat, layers_per_block = num_conv
    int vae = 0;                       // AutoencoderKL half: 1 = post_quant_conv + Decoder, 2 = Encoder + quant_conv (no time embedding,
                                       // no transformers; in_ch / out_ch are the half's own input / output channels, both <= 8)
    int dtype, in_ch, out_ch, n_levels, layers_per_block, ctx_dim, groups, linear_proj;
    float eps;
    int ch[MAX_LEVELS], attn[MAX_LEVELS], heads[MAX_LEVELS], tlayers[MAX_LEVELS];
    int temb_dim() const { return ch[0] * 4; }
};

struct Param {   // one engine-owned packed tensor (or a slice view of one)
    size_t off = 0;       // byte offset in the weight slab
    size_t bytes = 0;
    bool f32 = false;
};

enum OpClass { OC_CONV = 0, OC_LINEAR = 1, OC_ATTN = 2, OC_NORM = 3, OC_OTHER = 4, OC_COUNT = 5 };

struct Ref {
    enum Kind { NUL, WS, WT, SAMPLE, TIMESTEPS, CTX, OUT, DOWNRES, MIDRES, REFSTORE, CNCOND, CNOUT } kind = NUL;
    size_t off = 0;
    int idx = 0;
};

struct Run {
    unsigned char* ws; unsigned char* wt;
    const void* sample; const float* timesteps; const void* ctx; void* out;
    const void* const* down_res; const void* mid_res;
    unsigned char* ref_store;
    const void* cn_cond = nullptr;            // ControlNet: conditioning image [B, cond_ch, 8H, 8W] NCHW (io dtype)
    void* const* cn_out = nullptr;            // ControlNet: n_skips + 1 output tensors (NHWC, engine dtype)
    float cn_scale = 1.0f;                    // conditioning_scale
    int cn_accum = 0;                         // 1: add to what the outputs already hold (MultiControlNetModel's sum)
    hipStream_t stream;
    void* p(const Ref& r) const {
        switch (r.kind) {
            case Ref::WS: return ws + r.off;
            case Ref::WT: return wt + r.off;
            case Ref::SAMPLE: return (void*)((const unsigned char*)sample + r.off);
            case Ref::TIMESTEPS: return (void*)timesteps;
            case Ref::CTX: return (void*)((const unsigned char*)ctx + r.off);
            case Ref::OUT: return out;
            case Ref::DOWNRES: return (void*)down_res[r.idx];
            case Ref::MIDRES: return (void*)mid_res;
            case Ref::REFSTORE: return ref_store + r.off;
            case Ref::CNCOND: return (void*)cn_cond;
            case Ref::CNOUT: return cn_out[r.idx];
            default: return nullptr;
        }
    }
};

struct Op {
    int cls;
    double flops;
    const char* what;
    std::function<int(const Run&)> fn;
};

// attention-processor options of the reference that change the op list
//   ip_tokens > 0 : IPAttnProcessor2_0 (lib/models/architecture/ip_adapter/attention_processor.py:301-396): the last ip_tokens rows
//                   of encoder_hidden_states go through to_k_ip/to_v_ip and a second softmax, added with ip_scale;
//   ref_mode      : ReferenceAttnProc / ReferenceOnlyAttnProc (lib/models/architecture/diffusers.py:646-673,
//                   lib/pipelines/zero123plus.py:43-77): 1 = 'w' (store the self-attention keys/values of every layer),
//                   2 = 'r'/'m' (append the stored tokens to every self-attention's keys/values); ref_skip leading batch items
//                   neither store nor read (is_cfg_guidance); ref_H x ref_W = latent size of the pass that wrote the store.
struct AttnOpts {
    int ip_tokens = 0; float ip_scale = 1.0f;
    int ref_mode = 0, ref_H = 0, ref_W = 0, ref_skip = 0;
    bool operator==(const AttnOpts& o) const {
        return ip_tokens == o.ip_tokens && ip_scale == o.ip_scale && ref_mode == o.ref_mode && ref_H == o.ref_H && ref_W == o.ref_W &&
               ref_skip == o.ref_skip;
    }
};

struct Plan {
    int B = 0, H = 0, W = 0, n_img = 1, has_res = 0, io_dtype = 0, res_nhwc = 0, ctx_len = 0;
    AttnOpts ao;
    size_t ref_store_bytes = 0;
    unsigned long long last_use = 0;
    std::vector<Op> ops;
    size_t enc_end = 0;        // ops[0, enc_end) = unet_enc
    size_t ws_bytes = 0;
    double flops[OC_COUNT] = {0, 0, 0, 0, 0};
};

// plan-time first-fit allocator with coalescing
struct Arena {
    struct Blk { size_t off, size; bool free; };
    std::vector<Blk> b;
    size_t top = 0, peak = 0;
    size_t alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        for (size_t i = 0; i < b.size(); ++i)
            if (b[i].free && b[i].size >= bytes) {
                if (b[i].size > bytes) {
                    Blk rest{b[i].off + bytes, b[i].size - bytes, true};
                    b[i].size = bytes;
                    b.insert(b.begin() + i + 1, rest);
                }
                b[i].free = false;
                return b[i].off;
            }
        if (!b.empty() && b.back().free) {   // grow the trailing free block
            top += bytes - b.back().size;
            b.back().size = bytes;
            b.back().free = false;
            peak = top > peak ? top : peak;
            return b.back().off;
        }
        b.push_back({top, bytes, false});
        top += bytes;
        peak = top > peak ? top : peak;
        return b.back().off;
    }
    void release(size_t off) {
        for (size_t i = 0; i < b.size(); ++i)
            if (b[i].off == off && !b[i].free) {
                b[i].free = true;
                if (i + 1 < b.size() && b[i + 1].free) { b[i].size += b[i + 1].size; b.erase(b.begin() + i + 1); }
                if (i > 0 && b[i - 1].free) { b[i - 1].size += b[i].size; b.erase(b.begin() + i); }
                return;
            }
    }
};

struct Unet {
    Config cfg;
    std::map<std::string, Param> params;     // packed tensors, by engine name
    std::map<std::string, bool> loaded;      // diffusers names seen
    std::vector<std::string> expected;       // diffusers names required
    unsigned char* slab = nullptr;
    size_t slab_bytes = 0;
    int sum_temb = 0, sum_kv = 0;
    std::map<std::string, int> temb_off, kv_off;   // resnet prefix -> column offset; attn2 prefix -> column offset
    std::vector<std::unique_ptr<Plan>> plans;   // small LRU cache: 2-pass mode alternates write/read/decode plans every step
    Plan* cur = nullptr;
    unsigned long long tick = 0;
    AttnOpts ao;
    unsigned char* ref_store = nullptr;
    size_t ref_store_bytes = 0;
    int n_ip_loaded = 0, n_xf_layers = 0;
    bool fuse_sc = false;                       // conv_shortcut folded into conv2's K loop (all widths multiples of 64)
    std::map<std::string, int> sc_cin;          // resnet prefix -> input width, for resnets with a shortcut
    std::string err;
};

int esz(int dtype) { return dtype == MVE_F32 ? 4 : 2; }

// torchvision VGG16 `features` indices of the 13 convolutions, their widths, and lpips' five slices (relu1_2 ... relu5_3)
constexpr int VGG_IDX[13] = {0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28};
constexpr int VGG_CIN[13] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
constexpr int VGG_COUT[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
constexpr int VGG_SLICE[13] = {1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5};
constexpr int VGG_BLK_FIRST[6] = {0, 2, 4, 7, 10, 13};
std::string vgg_name(int i) { return "net.slice" + std::to_string(VGG_SLICE[i]) + "." + std::to_string(VGG_IDX[i]); }

// enumerate blocks in execution order ------------------------------------------------------------------
struct ResnetDesc { std::string name; int cin, cout; };
struct XfDesc { std::string name; int c, heads, layers; };

void enumerate(const Config& c, std::vector<ResnetDesc>& rs, std::vector<XfDesc>& xs) {
    const int n = c.n_levels, L = c.layers_per_block;
    if (c.sr || c.lpips) return;     // plain conv stacks
    if (c.vae) {      // diffusers Encoder / Decoder (autoencoders/vae.py): resnets only, one attention in the mid block
        const int Cm = c.ch[n - 1];
        if (c.vae == 2) {
            int cin = c.ch[0];
            for (int i = 0; i < n; ++i) {
                for (int j = 0; j < L; ++j)
                    rs.push_back({"down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? cin : c.ch[i], c.ch[i]});
                cin = c.ch[i];
            }
        }
        rs.push_back({"mid_block.resnets.0", Cm, Cm});
        rs.push_back({"mid_block.resnets.1", Cm, Cm});
        if (c.vae == 1) {
            int cin = Cm;
            for (int i = 0; i < n; ++i) {
                const int cout = c.ch[n - 1 - i];
                for (int j = 0; j < L + 1; ++j)
                    rs.push_back({"up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? cin : cout, cout});
                cin = cout;
            }
        }
        return;
    }
    int cin = c.ch[0];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < L; ++j) {
            rs.push_back({"down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? cin : c.ch[i], c.ch[i]});
            if (c.attn[i]) xs.push_back({"down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), c.ch[i], c.heads[i], c.tlayers[i]});
        }
        cin = c.ch[i];
    }
    rs.push_back({"mid_block.resnets.0", c.ch[n - 1], c.ch[n - 1]});
    xs.push_back({"mid_block.attentions.0", c.ch[n - 1], c.heads[n - 1], c.tlayers[n - 1]});
    rs.push_back({"mid_block.resnets.1", c.ch[n - 1], c.ch[n - 1]});
    if (c.controlnet) return;
    int prev = c.ch[n - 1];
    for (int i = 0; i < n; ++i) {
        const int lvl = n - 1 - i, cout = c.ch[lvl];
        const int in_blk = c.ch[(lvl - 1) > 0 ? (lvl - 1) : 0];
        for (int j = 0; j < L + 1; ++j) {
            const int skip = (j == L) ? in_blk : cout;
            const int rin = (j == 0) ? prev : cout;
            rs.push_back({"up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), rin + skip, cout});
            if (c.attn[lvl]) xs.push_back({"up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), cout, c.heads[lvl], c.tlayers[lvl]});
        }
        prev = cout;
    }
}

// ---------------------------------------------------------------------------------------------------
// parameter layout: reserve slab space for every packed tensor
// ---------------------------------------------------------------------------------------------------
struct SlabBuilder {
    Unet& u;
    size_t top = 0;
    void add(const std::string& name, size_t elems, bool f32) {
        Param p;
        p.off = top; p.f32 = f32; p.bytes = elems * (f32 ? 4 : 2);
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
            sb.add(b + ".qkv.w", 3 * C * C, false);
            need(b + ".attn1.to_q.weight"); need(b + ".attn1.to_k.weight"); need(b + ".attn1.to_v.weight");
            sb.add(b + ".o1.w", C * C, false); need(b + ".attn1.to_out.0.weight");
            sb.add(b + ".o1.b", C, true); need(b + ".attn1.to_out.0.bias");
            sb.add(b + ".q2.w", C * C, false); need(b + ".attn2.to_q.weight");
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
    auto mat = [&](Param* p, size_t elem_off, long long N, long long K, long long dst_row_stride) -> int {   // [N][K] rows
        MVE_CHECK(p, MVE_ERR_ARG, "load_param: no slot for %s", name.c_str());
        MVE_CHECK(numel() == N * K, MVE_ERR_ARG, "load_param(%s): expected %lldx%lld, got %lld elements", name.c_str(), N, K, numel());
        d = PackDims{{1, 1, N, K}, {0, 0, K, 1}, {0, 0, dst_row_stride, 1}, K};
        return pack(src_dtype, c.dtype, src, dstp(p, elem_off), d, s);
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
        rc = mat(P(b + ".qkv.w"), (size_t)which * C * C, C, C, C);
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
    } else if (ends_with(name, ".attn2.to_q.weight")) rc = mat(P(strip(name, ".attn2.to_q.weight") + ".q2.w"), 0, shape[0], shape[1], shape[1]);
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
    void rel(const Ref& r) { if (r.kind == Ref::WS) ar.release(r.off); }
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
        op(OC_LINEAR, 2.0 * M * N * K, what, [=](const Run& r) {
            return mve_gemm(d, r.p(A), lda, r.p(W), ldw, r.p(out), ldc, M, N, K, (const float*)r.p(bias), (const float*)r.p(rowvec),
                            ldrv, rpv, r.p(res), ldr, flags, out_scale, r.p(sk), skb, rimg, r.stream);
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
        op(OC_CONV, 2.0 * Bn * Ho * Wo * (double)Cout * 9 * C1, what, [=](const Run& r) {
            return mve_conv3x3(d, r.p(x), C1, nullptr, 0, Bn, H, W, stride, ups, r.p(Wt), Cout, r.p(out), Cout,
                               (const float*)r.p(bias), (const float*)r.p(rowvec), ldrv, r.p(res), Cout, fl, 1.0f, r.p(sk), skb, r.stream);
        });
        rel(sk);
    }
    void gn(Ref x1, int C1, Ref x2, int C2, int Bn, int HW, float eps, Ref g, Ref b, int silu, Ref out, const char* what) {
        const int d = dt, G = c.groups;
        const size_t wsb = mve_groupnorm_workspace_bytes(Bn, HW, C1 + C2, G);
        Ref scratch = ws(wsb);
        live(x1, what); live(x2, what); live(out, what);
        op(OC_NORM, 0, what, [=](const Run& r) {
            return mve_groupnorm_silu(d, r.p(x1), C1, r.p(x2), C2, Bn, HW, G, eps, (const float*)r.p(g), (const float*)r.p(b), silu,
                                      r.p(out), r.p(scratch), r.stream);
        });
        rel(scratch);
    }
    void ln(Ref x, Ref y, int M, int C, Ref g, Ref b) {
        const int d = dt;
        live(x, "layernorm"); live(y, "layernorm");
        op(OC_NORM, 0, "layernorm", [=](const Run& r) {
            return mve_layernorm(d, r.p(x), C, r.p(y), C, M, C, (const float*)r.p(g), (const float*)r.p(b), 1e-5f, r.stream);
        });
    }
    void attn(Ref q, int ldq, Ref k, int ldk, Ref v, int ldv, Ref o, int ldo, int Bn, int Lq, int Lk, int heads, int hd,
              Ref k2 = Ref(), int ldk2 = 0, Ref v2 = Ref(), int ldv2 = 0, int Lk2 = 0, const char* what = "attention") {
        const int d = dt;
        if (Bn <= 0) return;
        live(q, what); live(k, what); live(v, what); live(o, what);
        op(OC_ATTN, 4.0 * Bn * heads * (double)Lq * (Lk + Lk2) * hd, what, [=](const Run& r) {
            return mve_attention(d, r.p(q), ldq, r.p(k), ldk, r.p(v), ldv, r.p(k2), ldk2, r.p(v2), ldv2, r.p(o), ldo, Bn, Lq, Lk, Lk2, heads,
                                 hd, 1.0f / sqrtf((float)hd), r.stream);
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
            Ref out = ws((size_t)M * Cout * e);
            const int d = dt, Bn = B;
            Ref Wt = wt(name + ".conv2.w"), b2 = wt(name + ".conv2.b"), bs = wt(name + ".sc.b");
            live(h2, "resnet.conv2+shortcut"); live(x, "resnet.conv2+shortcut"); live(skip, "resnet.conv2+shortcut");
            const size_t skb = mve_gemm_workspace_bytes(M, Cout, 9 * Cout + Cin, H * W);
            Ref sk = skb ? ws(skb) : Ref();
            op(OC_CONV, 2.0 * M * (double)Cout * (9 * Cout + Cin), "resnet.conv2+shortcut", [=](const Run& r) {
                return mve_conv3x3_shortcut(d, r.p(h2), Cout, r.p(x), C1, r.p(skip), C2, Bn, H, W, r.p(Wt), Cout, r.p(out), Cout,
                                            (const float*)r.p(b2), (const float*)r.p(bs), nullptr, 0, 0, 1.0f, r.p(sk), skb, r.stream);
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
        Ref out = ws((size_t)M * Cout * e);
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
        Ref h = ws((size_t)M * C * e);
        gemm(n0, C, wt(name + ".proj_in.w"), C, h, C, M, C, C, wt(name + ".proj_in.b"), Ref(), 0, 0, Ref(), 0, 0, "transformer.proj_in");
        rel(n0);
        for (int k = 0; k < layers; ++k) {
            const std::string b = name + ".transformer_blocks." + std::to_string(k);
            // self attention
            Ref n1 = ws((size_t)M * C * e);
            ln(h, n1, M, C, wt(b + ".norm1.g"), wt(b + ".norm1.b"));
            Ref qkv = ws((size_t)M * 3 * C * e);
            gemm(n1, C, wt(b + ".qkv.w"), C, qkv, 3 * C, M, 3 * C, C, Ref(), Ref(), 0, 0, Ref(), 0, 0, "attn1.qkv");
            rel(n1);
            Ref a = ws((size_t)M * C * e);
            const AttnOpts& ao = pl.ao;
            if (ao.ref_mode == 0) {
                attn(qkv, 3 * C, at(qkv, (size_t)C * e), 3 * C, at(qkv, (size_t)2 * C * e), 3 * C, a, C, nb, L, L, heads, hd);
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
                    attn(qkv, 3 * C, at(qkv, (size_t)C * e), 3 * C, at(qkv, (size_t)2 * C * e), 3 * C, a, C, nb, L, L, heads, hd);
                    copy2d(st, (size_t)2 * C * e, at(qkv, ((size_t)skip * L * 3 * C + C) * e), (size_t)3 * C * e, (size_t)2 * C * e,
                           (size_t)nr * L, "reference K,V -> store");
                } else {
                    attn(qkv, 3 * C, at(qkv, (size_t)C * e), 3 * C, at(qkv, (size_t)2 * C * e), 3 * C, a, C, skip, L, L, heads, hd);
                    const size_t o = (size_t)skip * L * 3 * C * e;
                    attn(at(qkv, o), 3 * C, at(qkv, o + (size_t)C * e), 3 * C, at(qkv, o + (size_t)2 * C * e), 3 * C,
                         at(a, (size_t)skip * L * C * e), C, nr, L, L, heads, hd, st, 2 * C, at(st, (size_t)C * e), 2 * C, Lref,
                         "attention (+reference tokens)");
                }
            }
            rel(qkv);
            Ref h2 = ws((size_t)M * C * e);
            gemm(a, C, wt(b + ".o1.w"), C, h2, C, M, C, C, wt(b + ".o1.b"), Ref(), 0, 0, h, C, 0, "attn1.to_out+residual");
            rel(a); rel(h); h = h2;
            // cross attention (K/V hoisted)
            Ref n2 = ws((size_t)M * C * e);
            ln(h, n2, M, C, wt(b + ".norm2.g"), wt(b + ".norm2.b"));
            Ref q = ws((size_t)M * C * e);
            gemm(n2, C, wt(b + ".q2.w"), C, q, C, M, C, C, Ref(), Ref(), 0, 0, Ref(), 0, 0, "attn2.to_q");
            rel(n2);
            Ref a2 = ws((size_t)M * C * e);
            const size_t ko = (size_t)u.kv_off[b] * e;
            attn(q, C, at(ctxkv, ko), ld_kv, at(ctxkv, ko + (size_t)C * e), ld_kv, a2, C, nb, L, Lt, heads, hd);
            if (ao.ip_tokens > 0) {     // hidden_states + scale * SDPA(q, to_k_ip(ip), to_v_ip(ip))  (attention_processor.py:366-383)
                Ref aip = ws((size_t)M * C * e);
                attn(q, C, at(ipkv, ko), ld_kv, at(ipkv, ko + (size_t)C * e), ld_kv, aip, C, nb, L, ao.ip_tokens, heads, hd, Ref(), 0, Ref(), 0,
                     0, "attention (ip tokens)");
                const int d = dt;
                const float sc = ao.ip_scale;
                const size_t nel = (size_t)M * C;
                op(OC_OTHER, 0, "attn2 += scale * ip", [=](const Run& r) { return mve_axpy(d, r.p(a2), r.p(aip), sc, r.p(a2), nel, r.stream); });
                rel(aip);
            }
            rel(q);
            Ref h3 = ws((size_t)M * C * e);
            gemm(a2, C, wt(b + ".o2.w"), C, h3, C, M, C, C, wt(b + ".o2.b"), Ref(), 0, 0, h, C, 0, "attn2.to_out+residual");
            rel(a2); rel(h); h = h3;
            // feed forward (GEGLU fused in the first GEMM's epilogue)
            Ref n3 = ws((size_t)M * C * e);
            ln(h, n3, M, C, wt(b + ".norm3.g"), wt(b + ".norm3.b"));
            Ref f = ws((size_t)M * 4 * C * e);
            gemm(n3, C, wt(b + ".ff1.w"), C, f, 4 * C, M, 8 * C, C, wt(b + ".ff1.b"), Ref(), 0, 0, Ref(), 0, MVE_GEMM_GEGLU, "ff.geglu");
            rel(n3);
            Ref h4 = ws((size_t)M * C * e);
            gemm(f, 4 * C, wt(b + ".ff2.w"), 4 * C, h4, C, M, C, 4 * C, wt(b + ".ff2.b"), Ref(), 0, 0, h, C, 0, "ff.out+residual");
            rel(f); rel(h); h = h4;
        }
        Ref out = ws((size_t)M * C * e);
        gemm(h, C, wt(name + ".proj_out.w"), C, out, C, M, C, C, wt(name + ".proj_out.b"), Ref(), 0, 0, x, C, 0, "transformer.proj_out+residual");
        rel(h);
        return out;
    }

    // diffusers Attention of the VAE mid block (heads = 1, dim_head = C, residual_connection, bias everywhere, GroupNorm eps =
    // resnet eps): x + to_out(softmax(q k^T / sqrt(C)) v).  Head dim C = 512 is outside the fused attention kernel's range, so the
    // block runs on the GEMM kernel: scores (fp32) = q k^T, row softmax, P (V^T)^T with V^T produced directly by a GEMM whose "A"
    // operand is the weight matrix.  to_v's bias is added after P.V (rows of P sum to one).  Images are processed one after the
    // other through one [L, L] score buffer.
    Ref vae_attention(const std::string& name, Ref x, int C, int H, int W) {
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

    // LPIPS(net='vgg')(pred, target) and its gradient w.r.t. pred (lpips==0.1.4 as called from lib/models/losses/lpips_loss.py:8-42).
    // Forward ops = [0, enc_end): both images of every pair go through VGG16 as one batch of 2B; nothing is released, the
    // activations are the backward's inputs.  Backward ops = [enc_end, end): pred half only; every conv's dgrad is the forward conv
    // kernel on the transposed / flipped weight packing.
    int build_lpips(int B_, int H, int W, int io_dtype) {
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

    // SRVGGNetCompact.forward (lib/models/decoders/image_space_ss.py:63-70): conv + PReLU stack at the input resolution, last conv to
    // out_ch * r * r channels, PixelShuffle(r), plus the nearest-upsampled input.  H x W is the input size.
    int build_sr(int B_, int H, int W, int io_dtype) {
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

    // AutoencoderKL half (diffusers 0.27.2 autoencoders/vae.py Decoder / Encoder, as called at lib/pipelines/mvedit_3d_pipeline.py:1260
    // and :1441 of the reference).  H x W is the size of the half's INPUT (latent for the decoder, image for the encoder).
    int build_vae(int B_, int H, int W, int io_dtype) {
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
                    conv(x, cout, Bb, h, w, 1, 1, wt(un + ".w"), cout, y, wt(un + ".b"), Ref(), 0, Ref(), 0, "upsample+conv");
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

    int build(int B_, int H, int W, int n_img, int has_res, int io_dtype, int res_nhwc) {
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
        Ref x = ws((size_t)M0 * c.ch[0] * e);
        if (c.controlnet) {
            // controlnet_cond_embedding on the 8H x 8W conditioning image, added to conv_in(sample)
            const int Hc = 8 * H, Wc = 8 * W, cc = c.cond_ch;
            Ref cimg = ws((size_t)Bb * Hc * Wc * 8 * e);
            {
                Ref src; src.kind = Ref::CNCOND;
                op(OC_OTHER, 0, "cond nchw->nhwc", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(src), Bb, cc, Hc, Wc, 8, r.p(cimg), r.stream); });
            }
            const std::string en = "controlnet_cond_embedding.";
            int hc = Hc, wc = Wc, ci = 8;
            Ref cur = cimg;
            auto emb_conv = [&](const std::string& nm, int co, int stride, bool act) {
                const int ho = (hc - 1) / stride + 1, wo = (wc - 1) / stride + 1;
                Ref y = ws((size_t)Bb * ho * wo * co * e);
                conv(cur, ci, Bb, hc, wc, stride, 0, wt(nm + ".w"), co, y, wt(nm + ".b"), Ref(), 0, Ref(), 0, "cond_embedding.conv");
                rel(cur);
                if (act) {
                    const size_t nel = (size_t)Bb * ho * wo * co;
                    op(OC_OTHER, 0, "silu", [=](const Run& r) { return mve_silu(d, r.p(y), r.p(y), nel, r.stream); });
                }
                cur = y; hc = ho; wc = wo; ci = co;
            };
            emb_conv(en + "conv_in", CN_EMB[0], 1, true);
            for (int k = 0; k < 6; ++k) emb_conv(en + "blocks." + std::to_string(k), CN_EMB[(k + 1) / 2], (k & 1) ? 2 : 1, true);
            emb_conv(en + "conv_out", c.ch[0], 1, false);
            MVE_CHECK(hc == H && wc == W, MVE_ERR_ARG, "controlnet: conditioning image must be 8x the latent size");
            conv(x_in, 8, Bb, H, W, 1, 0, wt("conv_in.w"), c.ch[0], x, wt("conv_in.b"), Ref(), 0, cur, 0, "conv_in + cond_embedding");
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
                Ref y = ws((size_t)Bb * (h / 2) * (w / 2) * cin * e);
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
                Ref sum = ws(elems * e);
                Ref a = sk.r;
                const int C = sk.C, sh = sk.H, sw = sk.W;
                if (res_nhwc) {
                    op(OC_OTHER, 0, "skip += controlnet residual", [=](const Run& r) { return mve_axpy(d, r.p(a), r.p(src), 1.0f, r.p(sum), elems, r.stream); });
                } else {
                    Ref tmp = ws(elems * e);
                    op(OC_OTHER, 0, "residual nchw->nhwc", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(src), Bb, C, sh, sw, C, r.p(tmp), r.stream); });
                    op(OC_OTHER, 0, "skip += controlnet residual", [=](const Run& r) { return mve_axpy(d, r.p(a), r.p(tmp), 1.0f, r.p(sum), elems, r.stream); });
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
                Ref sum = ws(elems * e);
                const int hh = h, ww = w;
                if (res_nhwc) {
                    op(OC_OTHER, 0, "mid += controlnet residual", [=](const Run& r) { return mve_axpy(d, r.p(m), r.p(src), 1.0f, r.p(sum), elems, r.stream); });
                } else {
                    Ref tmp = ws(elems * e);
                    op(OC_OTHER, 0, "residual nchw->nhwc", [=](const Run& r) { return mve_nchw_to_nhwc(d, io_dtype, r.p(src), Bb, C, hh, ww, C, r.p(tmp), r.stream); });
                    op(OC_OTHER, 0, "mid += controlnet residual", [=](const Run& r) { return mve_axpy(d, r.p(m), r.p(tmp), 1.0f, r.p(sum), elems, r.stream); });
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
                Ref y = ws((size_t)Bb * (2 * h) * (2 * w) * cout * e);
                conv(x, cout, Bb, h, w, 1, 1, wt(un + ".w"), cout, y, wt(un + ".b"), Ref(), 0, Ref(), 0, "upsample+conv");
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
};

int ensure_plan(Unet& u, int B, int H, int W, int n_img, int has_res, int io_dtype, int res_nhwc, int ctx_len) {
    ++u.tick;
    for (auto& pp : u.plans) {
        const Plan& p = *pp;
        if (p.B == B && p.H == H && p.W == W && p.n_img == n_img && p.has_res == has_res && p.io_dtype == io_dtype &&
            p.res_nhwc == res_nhwc && p.ctx_len == ctx_len && p.ao == u.ao) {
            pp->last_use = u.tick;
            u.cur = pp.get();
            return MVE_OK;
        }
    }
    std::unique_ptr<Plan> np(new Plan());
    np->ao = u.ao;
    Builder b(u, *np);
    b.ctx_rows_per_img = ctx_len;
    u.cur = nullptr;
    const int rc = u.cfg.lpips ? b.build_lpips(B, H, W, io_dtype) : u.cfg.sr ? b.build_sr(B, H, W, io_dtype) : u.cfg.vae ? b.build_vae(B, H, W, io_dtype) : b.build(B, H, W, n_img, has_res, io_dtype, res_nhwc);
    if (rc != MVE_OK) return rc;
    np->ctx_len = ctx_len;
    np->last_use = u.tick;
    if (u.plans.size() >= 8) {
        size_t lru = 0;
        for (size_t i = 1; i < u.plans.size(); ++i)
            if (u.plans[i]->last_use < u.plans[lru]->last_use) lru = i;
        u.plans.erase(u.plans.begin() + lru);
    }
    u.plans.push_back(std::move(np));
    u.cur = u.plans.back().get();
    return MVE_OK;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int mve_unet_create(void** handle, int dtype, int in_channels, int out_channels, int n_levels, const int* block_out_channels,
                    int layers_per_block, const int* down_attn, const int* num_heads, const int* transformer_layers,
                    int cross_attention_dim, int norm_num_groups, float norm_eps, int use_linear_projection) {
    MVE_CHECK(handle, MVE_ERR_ARG, "unet_create: null handle");
    MVE_CHECK(dtype == MVE_F16 || dtype == MVE_BF16, MVE_ERR_ARG, "unet_create: dtype must be f16 or bf16");
    MVE_CHECK(n_levels >= 1 && n_levels <= MAX_LEVELS && layers_per_block >= 1, MVE_ERR_ARG, "unet_create: bad topology");
    MVE_CHECK(in_channels >= 1 && in_channels <= 8 && out_channels >= 1 && out_channels <= 8, MVE_ERR_ARG,
              "unet_create: in/out channels must be <= 8");
    MVE_CHECK(cross_attention_dim % 8 == 0, MVE_ERR_ARG, "unet_create: cross_attention_dim must be a multiple of 8");
    Unet* u = new Unet();
    Config& c = u->cfg;
    c.dtype = dtype; c.in_ch = in_channels; c.out_ch = out_channels; c.n_levels = n_levels; c.layers_per_block = layers_per_block;
    c.ctx_dim = cross_attention_dim; c.groups = norm_num_groups; c.eps = norm_eps; c.linear_proj = use_linear_projection;
    for (int i = 0; i < n_levels; ++i) {
        c.ch[i] = block_out_channels[i]; c.attn[i] = down_attn[i]; c.heads[i] = num_heads[i]; c.tlayers[i] = transformer_layers[i];
        if (c.ch[i] % 32 != 0 || c.ch[i] % c.groups != 0 || (c.attn[i] && c.ch[i] % c.heads[i] != 0)) {
            delete u;
            mve_set_error("unet_create: channel count %d incompatible with groups/heads", block_out_channels[i]);
            return MVE_ERR_ARG;
        }
    }
    layout_params(*u);   // host-side only; device storage is allocated by the first mve_unet_load_param
    *handle = u;
    return MVE_OK;
}

int mve_controlnet_create(void** handle, int dtype, int in_channels, int conditioning_channels, int n_levels, const int* block_out_channels,
                          int layers_per_block, const int* down_attn, const int* num_heads, const int* transformer_layers,
                          int cross_attention_dim, int norm_num_groups, float norm_eps, int use_linear_projection) {
    MVE_CHECK(conditioning_channels >= 1 && conditioning_channels <= 8, MVE_ERR_ARG, "controlnet_create: conditioning channels must be <= 8");
    // same topology arguments as the UNet; build the parameter table in ControlNet mode
    MVE_CHECK(handle, MVE_ERR_ARG, "controlnet_create: null handle");
    MVE_CHECK(dtype == MVE_F16 || dtype == MVE_BF16, MVE_ERR_ARG, "controlnet_create: dtype must be f16 or bf16");
    MVE_CHECK(n_levels >= 1 && n_levels <= MAX_LEVELS && layers_per_block >= 1, MVE_ERR_ARG, "controlnet_create: bad topology");
    MVE_CHECK(in_channels >= 1 && in_channels <= 8, MVE_ERR_ARG, "controlnet_create: in channels must be <= 8");
    MVE_CHECK(cross_attention_dim % 8 == 0, MVE_ERR_ARG, "controlnet_create: cross_attention_dim must be a multiple of 8");
    Unet* u = new Unet();
    Config& c = u->cfg;
    c.controlnet = 1; c.cond_ch = conditioning_channels;
    c.dtype = dtype; c.in_ch = in_channels; c.out_ch = in_channels; c.n_levels = n_levels; c.layers_per_block = layers_per_block;
    c.ctx_dim = cross_attention_dim; c.groups = norm_num_groups; c.eps = norm_eps; c.linear_proj = use_linear_projection;
    for (int i = 0; i < n_levels; ++i) {
        c.ch[i] = block_out_channels[i]; c.attn[i] = down_attn[i]; c.heads[i] = num_heads[i]; c.tlayers[i] = transformer_layers[i];
        if (c.ch[i] % 32 != 0 || c.ch[i] % c.groups != 0 || (c.attn[i] && c.ch[i] % c.heads[i] != 0)) {
            delete u;
            mve_set_error("controlnet_create: channel count %d incompatible with groups/heads", block_out_channels[i]);
            return MVE_ERR_ARG;
        }
    }
    layout_params(*u);
    *handle = u;
    return MVE_OK;
}

int mve_controlnet_forward(void* handle, const void* d_sample, int io_dtype, const float* d_timesteps, const void* d_ctx, const void* d_cond,
                           int B, int H, int W, int ctx_len, float conditioning_scale, int accumulate, void* const* d_outputs,
                           void* d_workspace, size_t workspace_bytes, float* op_ms, void* stream) {
    MVE_CHECK(handle, MVE_ERR_ARG, "controlnet_forward: null handle");
    Unet* u = (Unet*)handle;
    MVE_CHECK(u->cfg.controlnet, MVE_ERR_ARG, "controlnet_forward: handle is a UNet, not a ControlNet");
    {
        char first[256];
        const int miss = mve_unet_missing_params(handle, first, sizeof(first));
        MVE_CHECK(miss == 0, MVE_ERR_STATE, "controlnet_forward: %d parameters not loaded (first: %s)", miss, first);
    }
    MVE_CHECK(d_sample && d_timesteps && d_ctx && d_cond && d_outputs, MVE_ERR_ARG, "controlnet_forward: null pointer");
    int rc = ensure_plan(*u, B, H, W, 1, 0, io_dtype, 0, ctx_len);
    if (rc) return rc;
    const Plan& pl = *u->cur;
    MVE_CHECK(d_workspace && workspace_bytes >= pl.ws_bytes, MVE_ERR_NOMEM, "controlnet_forward: workspace %zu < required %zu", workspace_bytes,
              pl.ws_bytes);
    const int n_out = u->cfg.n_levels * (u->cfg.layers_per_block + 1) + 1;
    for (int i = 0; i < n_out; ++i) MVE_CHECK(d_outputs[i], MVE_ERR_ARG, "controlnet_forward: null output %d", i);
    Run r;
    r.ws = (unsigned char*)d_workspace; r.wt = u->slab;
    r.sample = d_sample; r.timesteps = d_timesteps; r.ctx = d_ctx; r.out = nullptr;
    r.down_res = nullptr; r.mid_res = nullptr; r.ref_store = nullptr;
    r.cn_cond = d_cond; r.cn_out = d_outputs; r.cn_scale = conditioning_scale; r.cn_accum = accumulate ? 1 : 0;
    r.stream = (hipStream_t)stream;
    std::vector<hipEvent_t> ev;
    if (op_ms) {
        ev.resize(pl.ops.size() + 1);
        for (auto& e : ev) MVE_HIP(hipEventCreate(&e));
        MVE_HIP(hipEventRecord(ev[0], r.stream));
    }
    for (size_t i = 0; i < pl.ops.size(); ++i) {
        rc = pl.ops[i].fn(r);
        if (rc) return rc;
        if (op_ms) MVE_HIP(hipEventRecord(ev[i + 1], r.stream));
    }
    if (op_ms) {
        MVE_HIP(hipStreamSynchronize(r.stream));
        for (size_t i = 0; i < pl.ops.size(); ++i) MVE_HIP(hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]));
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return MVE_OK;
}

int mve_vae_create(void** handle, int dtype, int half, int in_channels, int out_channels, int n_levels, const int* block_out_channels,
                   int layers_per_block, int norm_num_groups, float norm_eps) {
    MVE_CHECK(handle, MVE_ERR_ARG, "vae_create: null handle");
    MVE_CHECK(dtype == MVE_F16 || dtype == MVE_BF16, MVE_ERR_ARG, "vae_create: dtype must be f16 or bf16");
    MVE_CHECK(half == 1 || half == 2, MVE_ERR_ARG, "vae_create: half must be 1 (decoder) or 2 (encoder)");
    MVE_CHECK(n_levels >= 1 && n_levels <= MAX_LEVELS && layers_per_block >= 1, MVE_ERR_ARG, "vae_create: bad topology");
    MVE_CHECK(in_channels >= 1 && in_channels <= 8 && out_channels >= 1 && out_channels <= 8, MVE_ERR_ARG,
              "vae_create: in/out channels must be <= 8 (encoder: out_channels = 2 * latent_channels)");
    Unet* u = new Unet();
    Config& c = u->cfg;
    c.vae = half;
    c.dtype = dtype; c.in_ch = in_channels; c.out_ch = out_channels; c.n_levels = n_levels; c.layers_per_block = layers_per_block;
    c.ctx_dim = 8; c.groups = norm_num_groups; c.eps = norm_eps; c.linear_proj = 0;
    for (int i = 0; i < n_levels; ++i) {
        c.ch[i] = block_out_channels[i]; c.attn[i] = 0; c.heads[i] = 1; c.tlayers[i] = 0;
        if (c.ch[i] % 8 != 0 || c.groups <= 0 || c.ch[i] % c.groups != 0) {
            delete u;
            mve_set_error("vae_create: channel count %d incompatible with %d groups", block_out_channels[i], norm_num_groups);
            return MVE_ERR_ARG;
        }
    }
    layout_params(*u);
    *handle = u;
    return MVE_OK;
}

// image networks (VAE halves, SRVGGNetCompact): one NCHW tensor in, one out, no conditioning
static int imgnet_plan(void* handle, bool want_vae, int B, int H, int W, int io_dtype, size_t* workspace_bytes, int* n_ops, double* flops) {
    MVE_CHECK(handle && B > 0 && H > 0 && W > 0, MVE_ERR_ARG, "plan: bad arguments");
    Unet* u = (Unet*)handle;
    MVE_CHECK(want_vae ? u->cfg.vae != 0 : u->cfg.sr != 0, MVE_ERR_ARG, "plan: handle is not %s", want_vae ? "a VAE half" : "an SRVGGNetCompact");
    int rc = ensure_plan(*u, B, H, W, 1, 0, io_dtype, 0, 0);
    if (rc) return rc;
    if (workspace_bytes) *workspace_bytes = u->cur->ws_bytes;
    if (n_ops) *n_ops = (int)u->cur->ops.size();
    if (flops) for (int i = 0; i < OC_COUNT; ++i) flops[i] = u->cur->flops[i];
    return MVE_OK;
}

static int imgnet_forward(void* handle, bool want_vae, const void* d_in, int io_dtype, int B, int H, int W, void* d_out, void* d_workspace,
                          size_t workspace_bytes, float* op_ms, void* stream) {
    MVE_CHECK(handle, MVE_ERR_ARG, "forward: null handle");
    Unet* u = (Unet*)handle;
    MVE_CHECK(want_vae ? u->cfg.vae != 0 : u->cfg.sr != 0, MVE_ERR_ARG, "forward: handle is not %s", want_vae ? "a VAE half" : "an SRVGGNetCompact");
    {
        char first[256];
        const int miss = mve_unet_missing_params(handle, first, sizeof(first));
        MVE_CHECK(miss == 0, MVE_ERR_STATE, "forward: %d parameters not loaded (first: %s)", miss, first);
    }
    MVE_CHECK(d_in && d_out, MVE_ERR_ARG, "forward: null pointer");
    int rc = ensure_plan(*u, B, H, W, 1, 0, io_dtype, 0, 0);
    if (rc) return rc;
    const Plan& pl = *u->cur;
    MVE_CHECK(d_workspace && workspace_bytes >= pl.ws_bytes, MVE_ERR_NOMEM, "forward: workspace %zu < required %zu", workspace_bytes,
              pl.ws_bytes);
    Run r;
    r.ws = (unsigned char*)d_workspace; r.wt = u->slab;
    r.sample = d_in; r.timesteps = nullptr; r.ctx = nullptr; r.out = d_out;
    r.down_res = nullptr; r.mid_res = nullptr; r.ref_store = nullptr;
    r.stream = (hipStream_t)stream;
    std::vector<hipEvent_t> ev;
    if (op_ms) {
        ev.resize(pl.ops.size() + 1);
        for (auto& e : ev) MVE_HIP(hipEventCreate(&e));
        MVE_HIP(hipEventRecord(ev[0], r.stream));
    }
    for (size_t i = 0; i < pl.ops.size(); ++i) {
        rc = pl.ops[i].fn(r);
        if (rc) return rc;
        if (op_ms) MVE_HIP(hipEventRecord(ev[i + 1], r.stream));
    }
    if (op_ms) {
        MVE_HIP(hipStreamSynchronize(r.stream));
        for (size_t i = 0; i < pl.ops.size(); ++i) MVE_HIP(hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]));
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return MVE_OK;
}

int mve_vae_plan(void* handle, int B, int H, int W, int io_dtype, size_t* workspace_bytes, int* n_ops, double* flops) {
    return imgnet_plan(handle, true, B, H, W, io_dtype, workspace_bytes, n_ops, flops);
}
int mve_vae_forward(void* handle, const void* d_in, int io_dtype, int B, int H, int W, void* d_out, void* d_workspace,
                    size_t workspace_bytes, float* op_ms, void* stream) {
    return imgnet_forward(handle, true, d_in, io_dtype, B, H, W, d_out, d_workspace, workspace_bytes, op_ms, stream);
}

int mve_srvgg_create(void** handle, int dtype, int num_in_ch, int num_out_ch, int num_feat, int num_conv, int upscale) {
    MVE_CHECK(handle, MVE_ERR_ARG, "srvgg_create: null handle");
    MVE_CHECK(dtype == MVE_F16 || dtype == MVE_BF16, MVE_ERR_ARG, "srvgg_create: dtype must be f16 or bf16");
    MVE_CHECK(num_in_ch >= 1 && num_in_ch <= 8 && num_out_ch == num_in_ch, MVE_ERR_ARG,
              "srvgg_create: 1..8 channels, num_out_ch == num_in_ch (the input is added to the output, image_space_ss.py:68-69)");
    MVE_CHECK(num_feat >= 8 && num_feat % 8 == 0 && num_conv >= 0 && upscale >= 1 && upscale <= 8, MVE_ERR_ARG, "srvgg_create: bad topology");
    Unet* u = new Unet();
    Config& c = u->cfg;
    c.sr = 1; c.sr_scale = upscale;
    c.dtype = dtype; c.in_ch = num_in_ch; c.out_ch = num_out_ch; c.n_levels = 1; c.layers_per_block = num_conv;
    c.ctx_dim = 8; c.groups = 1; c.eps = 0.f; c.linear_proj = 0;
    c.ch[0] = num_feat; c.attn[0] = 0; c.heads[0] = 1; c.tlayers[0] = 0;
    layout_params(*u);
    *handle = u;
    return MVE_OK;
}
int mve_srvgg_plan(void* handle, int B, int H, int W, int io_dtype, size_t* workspace_bytes, int* n_ops, double* flops) {
    return imgnet_plan(handle, false, B, H, W, io_dtype, workspace_bytes, n_ops, flops);
}
int mve_srvgg_forward(void* handle, const void* d_in, int io_dtype, int B, int H, int W, void* d_out, void* d_workspace,
                      size_t workspace_bytes, float* op_ms, void* stream) {
    return imgnet_forward(handle, false, d_in, io_dtype, B, H, W, d_out, d_workspace, workspace_bytes, op_ms, stream);
}

int mve_lpips_create(void** handle, int dtype, int normalize_inputs) {
    MVE_CHECK(handle, MVE_ERR_ARG, "lpips_create: null handle");
    MVE_CHECK(dtype == MVE_F16 || dtype == MVE_BF16, MVE_ERR_ARG, "lpips_create: dtype must be f16 or bf16");
    Unet* u = new Unet();
    Config& c = u->cfg;
    c.lpips = 1; c.lpips_normalize = normalize_inputs ? 1 : 0;
    c.dtype = dtype; c.in_ch = 3; c.out_ch = 3; c.n_levels = 1; c.layers_per_block = 1;
    c.ctx_dim = 8; c.groups = 1; c.eps = 0.f; c.linear_proj = 0;
    c.ch[0] = 64; c.attn[0] = 0; c.heads[0] = 1; c.tlayers[0] = 0;
    layout_params(*u);
    *handle = u;
    return MVE_OK;
}

int mve_lpips_plan(void* handle, int B, int H, int W, int io_dtype, size_t* workspace_bytes, int* n_ops, int* n_forward_ops, double* flops) {
    MVE_CHECK(handle && B > 0 && H > 0 && W > 0, MVE_ERR_ARG, "lpips_plan: bad arguments");
    Unet* u = (Unet*)handle;
    MVE_CHECK(u->cfg.lpips, MVE_ERR_ARG, "lpips_plan: handle is not an LPIPS engine");
    int rc = ensure_plan(*u, B, H, W, 1, 0, io_dtype, 0, 0);
    if (rc) return rc;
    if (workspace_bytes) *workspace_bytes = u->cur->ws_bytes;
    if (n_ops) *n_ops = (int)u->cur->ops.size();
    if (n_forward_ops) *n_forward_ops = (int)u->cur->enc_end;
    if (flops) for (int i = 0; i < OC_COUNT; ++i) flops[i] = u->cur->flops[i];
    return MVE_OK;
}

// phase 1: loss[n] = LPIPS(pred[n], target[n]);  phase 2: grad_pred = d (sum_n grad_loss[n] * loss[n]) / d pred.  Phase 2 reads the
// activations phase 1 left in d_workspace: same workspace, same (B, H, W), no other call on it in between.
static int lpips_run(void* handle, int phase, const void* d_pred, const void* d_target, const float* d_grad_loss, int io_dtype, int B, int H,
                     int W, void* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
    MVE_CHECK(handle, MVE_ERR_ARG, "lpips: null handle");
    Unet* u = (Unet*)handle;
    MVE_CHECK(u->cfg.lpips, MVE_ERR_ARG, "lpips: handle is not an LPIPS engine");
    {
        char first[256];
        const int miss = mve_unet_missing_params(handle, first, sizeof(first));
        MVE_CHECK(miss == 0, MVE_ERR_STATE, "lpips: %d parameters not loaded (first: %s)", miss, first);
    }
    MVE_CHECK(d_out && (phase == 2 ? d_grad_loss != nullptr : (d_pred && d_target)), MVE_ERR_ARG, "lpips: null pointer");
    int rc = ensure_plan(*u, B, H, W, 1, 0, io_dtype, 0, 0);
    if (rc) return rc;
    const Plan& pl = *u->cur;
    MVE_CHECK(d_workspace && workspace_bytes >= pl.ws_bytes, MVE_ERR_NOMEM, "lpips: workspace %zu < required %zu", workspace_bytes, pl.ws_bytes);
    Run r;
    r.ws = (unsigned char*)d_workspace; r.wt = u->slab;
    r.sample = d_pred; r.timesteps = d_grad_loss; r.ctx = d_target; r.out = d_out;
    r.down_res = nullptr; r.mid_res = nullptr; r.ref_store = nullptr;
    r.stream = (hipStream_t)stream;
    const size_t lo = phase == 2 ? pl.enc_end : 0, hi = phase == 1 ? pl.enc_end : pl.ops.size();
    for (size_t i = lo; i < hi; ++i) {
        rc = pl.ops[i].fn(r);
        if (rc) return rc;
    }
    return MVE_OK;
}

int mve_lpips_forward(void* handle, const void* d_pred, const void* d_target, int io_dtype, int B, int H, int W, float* d_loss,
                      void* d_workspace, size_t workspace_bytes, void* stream) {
    return lpips_run(handle, 1, d_pred, d_target, nullptr, io_dtype, B, H, W, d_loss, d_workspace, workspace_bytes, stream);
}

int mve_lpips_backward(void* handle, const float* d_grad_loss, int io_dtype, int B, int H, int W, void* d_grad_pred, void* d_workspace,
                       size_t workspace_bytes, void* stream) {
    return lpips_run(handle, 2, nullptr, nullptr, d_grad_loss, io_dtype, B, H, W, d_grad_pred, d_workspace, workspace_bytes, stream);
}

int mve_unet_tune(int fuse_shortcut) {
    const int old = g_fuse_shortcut;
    if (fuse_shortcut >= 0) g_fuse_shortcut = fuse_shortcut ? 1 : 0;
    return old;
}

int mve_unet_destroy(void* handle) {
    if (!handle) return MVE_OK;
    Unet* u = (Unet*)handle;
    if (u->slab) (void)hipFree(u->slab);
    delete u;
    return MVE_OK;
}

size_t mve_unet_weight_bytes(void* handle) { return handle ? ((Unet*)handle)->slab_bytes : 0; }

int mve_unet_load_param(void* handle, const char* name, const void* d_src, int src_dtype, int ndim, const long long* shape,
                        void* stream) {
    MVE_CHECK(handle && name && d_src && shape && ndim >= 1 && ndim <= 4, MVE_ERR_ARG, "unet_load_param: bad arguments");
    Unet* u = (Unet*)handle;
    if (!u->slab) {
        hipError_t e = hipMalloc((void**)&u->slab, u->slab_bytes);
        if (e != hipSuccess) {
            u->slab = nullptr;
            mve_set_error("unet_load_param: hipMalloc(%zu) failed: %s", u->slab_bytes, hipGetErrorString(e));
            return MVE_ERR_HIP;
        }
        MVE_HIP(hipMemsetAsync(u->slab, 0, u->slab_bytes, (hipStream_t)stream));      // padding rows / optional weights start as zeros
    }
    std::string nm = name;
    if (u->cfg.lpips && nm.compare(0, 5, "lins.") == 0) nm = "lin" + nm.substr(5);      // nn.ModuleList alias of lin0..lin4
    if (u->cfg.vae) {     // AutoencoderKL state-dict names -> the half's own
        const std::string own = u->cfg.vae == 1 ? "decoder." : "encoder.", pq = u->cfg.vae == 1 ? "post_quant_conv." : "quant_conv.";
        if (nm.compare(0, own.size(), own) == 0) nm = nm.substr(own.size());
        else if (nm.compare(0, pq.size(), pq) == 0) nm = "pq_conv." + nm.substr(pq.size());
        else {
            mve_set_error("vae_load_param: %s does not belong to the %s", name, u->cfg.vae == 1 ? "decoder" : "encoder");
            return MVE_ERR_ARG;
        }
    }
    return load_param(*u, nm, d_src, src_dtype, ndim, shape, (hipStream_t)stream);
}

int mve_unet_missing_params(void* handle, char* buf, int buf_len) {
    MVE_CHECK(handle, MVE_ERR_ARG, "unet_missing_params: null handle");
    Unet* u = (Unet*)handle;
    int missing = 0;
    std::string first;
    for (auto& n : u->expected)
        if (!u->loaded.count(n)) { if (!missing) first = n; ++missing; }
    if (buf && buf_len > 0) { strncpy(buf, first.c_str(), (size_t)buf_len - 1); buf[buf_len - 1] = 0; }
    return missing;
}

int mve_unet_plan(void* handle, int B, int H, int W, int ctx_len, int num_cross_attn_imgs, int has_residuals, int io_dtype,
                  int residuals_nhwc, size_t* workspace_bytes, int* n_ops, double* flops /* [5]: conv, linear, attention, norm, other */) {
    MVE_CHECK(handle && B > 0 && H > 0 && W > 0 && ctx_len > 0 && num_cross_attn_imgs >= 1, MVE_ERR_ARG, "unet_plan: bad arguments");
    Unet* u = (Unet*)handle;
    int rc = ensure_plan(*u, B, H, W, num_cross_attn_imgs, has_residuals, io_dtype, residuals_nhwc, ctx_len);
    if (rc) return rc;
    if (workspace_bytes) *workspace_bytes = u->cur->ws_bytes;
    if (n_ops) *n_ops = (int)u->cur->ops.size();
    if (flops) for (int i = 0; i < OC_COUNT; ++i) flops[i] = u->cur->flops[i];
    return MVE_OK;
}

int mve_unet_forward(void* handle, int phase, const void* d_sample, int io_dtype, const float* d_timesteps, const void* d_ctx,
                     int B, int H, int W, int ctx_len, int num_cross_attn_imgs, const void* const* down_residuals,
                     const void* d_mid_residual, int residuals_nhwc, void* d_out, void* d_workspace, size_t workspace_bytes,
                     float* op_ms /* optional host array [n_ops]: per-op milliseconds (synchronises) */, void* stream) {
    MVE_CHECK(handle, MVE_ERR_ARG, "unet_forward: null handle");
    Unet* u = (Unet*)handle;
    MVE_CHECK(!u->cfg.controlnet && !u->cfg.vae && !u->cfg.sr && !u->cfg.lpips, MVE_ERR_ARG, "unet_forward: handle is a ControlNet / VAE / SRVGG (use their own forward calls)");
    {
        char first[256];
        const int miss = mve_unet_missing_params(handle, first, sizeof(first));
        MVE_CHECK(miss == 0, MVE_ERR_STATE, "unet_forward: %d parameters not loaded (first: %s)", miss, first);
    }
    const int has_res = (down_residuals && d_mid_residual) ? 1 : 0;
    int rc = ensure_plan(*u, B, H, W, num_cross_attn_imgs, has_res, io_dtype, residuals_nhwc, ctx_len);
    if (rc) return rc;
    const Plan& pl = *u->cur;
    MVE_CHECK(d_workspace && workspace_bytes >= pl.ws_bytes, MVE_ERR_NOMEM, "unet_forward: workspace %zu < required %zu",
              workspace_bytes, pl.ws_bytes);
    MVE_CHECK(d_timesteps && d_ctx && (phase == 1 || d_out) && (phase == 2 || d_sample), MVE_ERR_ARG, "unet_forward: null pointer");
    Run r;
    r.ws = (unsigned char*)d_workspace; r.wt = u->slab;
    r.sample = d_sample; r.timesteps = d_timesteps; r.ctx = d_ctx; r.out = d_out;
    r.down_res = down_residuals; r.mid_res = d_mid_residual;
    r.ref_store = u->ref_store;
    MVE_CHECK(pl.ref_store_bytes == 0 || (u->ref_store && u->ref_store_bytes >= pl.ref_store_bytes), MVE_ERR_NOMEM,
              "unet_forward: reference store %zu < required %zu bytes", u->ref_store_bytes, pl.ref_store_bytes);
    r.stream = (hipStream_t)stream;
    const size_t lo = phase == 2 ? pl.enc_end : 0, hi = phase == 1 ? pl.enc_end : pl.ops.size();
    std::vector<hipEvent_t> ev;
    if (op_ms) {
        ev.resize(hi - lo + 1);
        for (auto& e : ev) MVE_HIP(hipEventCreate(&e));
        MVE_HIP(hipEventRecord(ev[0], r.stream));
    }
    for (size_t i = lo; i < hi; ++i) {
        rc = pl.ops[i].fn(r);
        if (rc) return rc;
        if (op_ms) MVE_HIP(hipEventRecord(ev[i - lo + 1], r.stream));
    }
    if (op_ms) {
        MVE_HIP(hipStreamSynchronize(r.stream));
        for (size_t i = lo; i < hi; ++i) MVE_HIP(hipEventElapsedTime(&op_ms[i], ev[i - lo], ev[i - lo + 1]));
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return MVE_OK;
}

int mve_unet_set_attention(void* handle, int ip_tokens, float ip_scale, int ref_mode, int ref_H, int ref_W, int ref_skip,
                           void* d_ref_store, size_t ref_store_bytes) {
    MVE_CHECK(handle && ip_tokens >= 0 && ref_mode >= 0 && ref_mode <= 2 && ref_skip >= 0, MVE_ERR_ARG, "unet_set_attention: bad arguments");
    Unet* u = (Unet*)handle;
    u->ao.ip_tokens = ip_tokens; u->ao.ip_scale = ip_tokens ? ip_scale : 1.0f;
    u->ao.ref_mode = ref_mode; u->ao.ref_skip = ref_mode ? ref_skip : 0;
    u->ao.ref_H = ref_mode == 2 ? ref_H : 0; u->ao.ref_W = ref_mode == 2 ? ref_W : 0;
    u->ref_store = (unsigned char*)d_ref_store; u->ref_store_bytes = ref_store_bytes;
    return MVE_OK;
}

size_t mve_unet_ref_store_bytes(void* handle, int B, int ref_H, int ref_W, int ref_skip) {
    if (!handle || B <= ref_skip) return 0;
    const Config& c = ((Unet*)handle)->cfg;
    std::vector<ResnetDesc> rs;
    std::vector<XfDesc> xs;
    enumerate(c, rs, xs);
    size_t total = 0;
    for (auto& x : xs) {
        int lvl = 0;
        if (x.name.compare(0, 12, "down_blocks.") == 0) lvl = atoi(x.name.c_str() + 12);
        else if (x.name.compare(0, 10, "up_blocks.") == 0) lvl = c.n_levels - 1 - atoi(x.name.c_str() + 10);
        else lvl = c.n_levels - 1;
        const size_t L = (size_t)(ref_H >> lvl) * (ref_W >> lvl);
        total += (size_t)x.layers * (B - ref_skip) * L * 2 * x.c * 2;
    }
    return total;
}

/* describe op i of the current plan: class (0 conv,1 linear,2 attention,3 norm,4 other), flops, label */
int mve_unet_op_info(void* handle, int i, int* cls, double* flops, char* label, int label_len) {
    MVE_CHECK(handle, MVE_ERR_ARG, "unet_op_info: null handle");
    Unet* u = (Unet*)handle;
    MVE_CHECK(u->cur && i >= 0 && i < (int)u->cur->ops.size(), MVE_ERR_ARG, "unet_op_info: no such op %d", i);
    const Op& o = u->cur->ops[i];
    if (cls) *cls = o.cls;
    if (flops) *flops = o.flops;
    if (label && label_len > 0) { strncpy(label, o.what, (size_t)label_len - 1); label[label_len - 1] = 0; }
    return (i < (int)u->cur->enc_end) ? 1 : 2;   // which phase the op belongs to
}

}  // extern "C"
