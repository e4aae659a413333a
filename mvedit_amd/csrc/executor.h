// Fragment of the executor's single translation unit (csrc/unet.hip includes it; not a stand-alone header): primitive declarations, weight packing kernel, configuration, parameter table, plan / op / arena types, block enumeration.
#pragma once
#include "common.h"

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

extern "C" {
int mve_gemm(int, const void*, int, const void*, int, void*, int, int, int, int, const float*, const float*, int, int,
             const void*, int, int, float, void*, size_t, int, void*);
int mve_gemm_pair(int, const void*, int, const void*, int, void*, int, int, int, int, const float*, const float*, int, int,
                  const void*, int, int, float, void*, size_t, int, const void*, void*, void*);
int mve_gemm_pair_ln(int, const void*, int, const void*, int, void*, int, int, int, int, const float*, const void*, int, void*, size_t, int,
                     const void*, void*, void*, int, const float*, const float*, float, void*);
int mve_conv3x3_pair(int, const void*, int, const void*, int, int, int, int, int, int, const void*, int, void*, int,
                     const float*, const float*, int, const void*, int, int, float, void*, size_t, const void*, void*, void*);
int mve_conv3x3_shortcut_pair(int, const void*, int, const void*, int, const void*, int, int, int, int, const void*, int, void*, int,
                              const float*, const float*, int, float, void*, size_t, void*, void*);
int mve_groupnorm_silu_pair(int, const void*, int, const void*, int, int, int, int, float, const float*, const float*, int, void*,
                            void*, const void*, const void*, void*);
int mve_layernorm_pair(int, const void*, int, void*, int, int, int, const float*, const float*, float, const void*, void*);
int mve_axpy_pair(int, const void*, const void*, const void*, float, void*, void*, size_t, void*);
int mve_conv3x3(int, const void*, int, const void*, int, int, int, int, int, int, const void*, int, void*, int,
                const float*, const float*, int, const void*, int, int, float, void*, size_t, void*);
size_t mve_gemm_workspace_bytes(int, int, int, int);
int mve_upsample_conv_phases_supported(int, int, int, int, int);
size_t mve_upsample_conv_phases_workspace_bytes(int, int, int, int, int);
int mve_pack_upsample_phase_weights(int, int, const void*, int, int, void*, void*);
int mve_upsample_conv_phases(int, const void*, int, int, int, int, const void*, int, void*, const float*, int, void*, size_t, void*, void*);
int mve_attention_prescaled(int, const void*, int, const void*, int, const void*, int, const void*, int, const void*, int, void*, int, int, int, int, int, int, int, void*);
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
struct PackDims { long long D[4], s[4], t[4]; long long valid3; float mul = 1.0f; };   // mul: applied in fp32 before the single rounding

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
        if (d3 < p.valid3) v = (float)src[d0 * p.s[0] + d1 * p.s[1] + d2 * p.s[2] + d3 * p.s[3]] * p.mul;
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
    float q_fold = 0.f;   // > 0: a to_q weight whose rows carry head_dim^-1/2 * log2(e) (the attention kernel then takes Q as it comes)
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
//   residual_pair : not a processor option but a plan option of this engine (mve_unet_set_residual_mode): the residual stream of ResnetBlock2D /
//                   BasicTransformerBlock / Transformer2DModel is kept as an unrounded (hi, lo) pair of 16-bit tensors (include/mvedit_amd.h).
//   cn_cond_repeat: ControlNet plans only (mve_controlnet_set_cond_repeat): the conditioning images are given for B / R items and item b uses image
//                   b mod (B / R) -- under classifier-free guidance both halves of the batch see the same control images
//                   (mvedit_3d_pipeline.py:1232: `ctrl_images.split(diff_bs) * 2`), so the conditioning embedding runs once, not R times.
struct AttnOpts {
    int ip_tokens = 0; float ip_scale = 1.0f;
    int ref_mode = 0, ref_H = 0, ref_W = 0, ref_skip = 0;
    int residual_pair = 0;
    int cn_cond_repeat = 1;
    bool operator==(const AttnOpts& o) const {
        return ip_tokens == o.ip_tokens && ip_scale == o.ip_scale && ref_mode == o.ref_mode && ref_H == o.ref_H && ref_W == o.ref_W &&
               ref_skip == o.ref_skip && residual_pair == o.residual_pair && cn_cond_repeat == o.cn_cond_repeat;
    }
};

struct Plan {
    int B = 0, H = 0, W = 0, n_img = 1, has_res = 0, io_dtype = 0, res_nhwc = 0, ctx_len = 0;
    AttnOpts ao;
    size_t ref_store_bytes = 0;
    unsigned long long last_use = 0;
    unsigned long long uid = 0;       // unique per built plan (graph cache key: plan objects are recycled by the LRU)
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
    // optional hipGraph replay of mve_unet_forward (mve_unet_graph): a call whose plan AND every pointer argument equal an earlier
    // call's is captured on its second sighting and replayed from then on (launch-bound small-batch forwards, e.g. 8 images per rank)
    struct GraphEntry {
        unsigned long long plan_uid = 0;
        int phase = 0, seen = 0;
        std::vector<const void*> ptrs;          // sample, timesteps, ctx, out, workspace, mid residual, ref store, stream, down residuals...
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        unsigned long long last_use = 0;
    };
    bool graph_mode = false;
    std::vector<GraphEntry> graphs;
    unsigned long long plan_counter = 0;
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

}  // namespace
