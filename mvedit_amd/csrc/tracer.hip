// Kernels of the TRACER-B7 foreground segmentor (lib/models/segmentors/tracer_b7.py:16-73 over lib/models/architecture/tracerb7/: EfficientNet-B7
// encoder + TRACER decoder), the pieces that the GEMM / 3x3-conv kernels do not cover.  Activations are NHWC 16-bit ([B, H, W, C] contiguous, the
// layout every 1x1 convolution consumes as a plain [M, C] GEMM operand); single-channel decoder maps are fp32.  BatchNorm is folded into the
// convolution weights and a per-channel bias by the host mirror (mvedit_amd/segmentor.py) when the state dict is loaded.  The 1x1 / 1xk / kx1 /
// dilated 3x3 layers run on the matrix cores here (mve_seg_mconv: v_mfma_f32_16x16x32 with operands loaded straight into the instruction's
// layout, further down); dense 3x3 convolutions of the decoder go to mve_conv3x3; the rest are HBM- / L2-bound gathers with fp32 arithmetic:
//   mve_seg_conv2d        depthwise (groups = C) or small dense convolution: any kernel size / stride / dilation / explicit top-left padding,
//                         bias, activation, optional elementwise multiply / add of a second tensor, output into a channel slice of a wider tensor
//   mve_seg_act           in-place activation of a GEMM / conv3x3 output (swish, SELU, ReLU, sigmoid)
//   mve_seg_channel_mean  global average pool [B, HW, C] -> [B, C] fp32 (squeeze-and-excite, union attention)
//   mve_seg_se_gate       squeeze-and-excite gate: sigmoid(W2 swish(W1 pooled + b1) + b2), one block per image (efficientnet.py:124-129)
//   mve_seg_scale         x[b, p, c] = x[b, p, c] * A[b, c] + S[b, c] (SE gating; union attention's channel tracing + BatchNorm + confidence mask)
//   mve_seg_resize        bilinear resize (align_corners on / off, no antialias) of NHWC 16-bit or fp32 maps, optional mean / std normalisation
//                         and NCHW fp32 input (the wrapper's torchvision Resize + Normalize, tracer_b7.py:39-44)
//   mve_seg_uam_channel   union attention, channel tracer (att_modules.py:147-168): BatchNorm of the pooled vector, q / k / v / fc mat-vecs, the
//                         C x C softmax (SDPA with head dim 1, scale 1), sigmoid, and the 10 % quantile confidence mask (:135-145)
//   mve_seg_uam_spatial   union attention, spatial part (:176-187): softmax(q k^T) v + v over the rows of an H x W map
//   mve_seg_object_mix    object attention input (att_modules.py:277-282): enc * (sigmoid(d) + edge), edge = 1 - sigmoid(d) where that is <= 0.93
//   mve_seg_fuse          sigmoid of the mean of the three bilinearly upsampled side outputs (tracer.py:86-97)
//   mve_seg_post          3x3 min-pool erosion, resize to the caller's size, and the failure rule (tracer_b7.py:66-72)
#include "common.h"

#include <math.h>

namespace {

enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_SELU = 2, ACT_RELU = 3, ACT_SIGMOID = 4 };

__device__ __forceinline__ float seg_act(float v, int act) {
    switch (act) {
        case ACT_SWISH: return v / (1.0f + __expf(-v));
        case ACT_SELU: return 1.0507009873554805f * (v > 0.f ? v : 1.6732632423543772f * (__expf(v) - 1.0f));
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

template <class Tag>
__device__ __forceinline__ void load8f(const typename Tag::T* p, float (&v)[8]) {
    const typename Tag::V8 t = *reinterpret_cast<const typename Tag::V8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = Tag::to_f32(t[e]);
}

struct ConvD {
    const void* x; const float* w; const float* bias; void* out; const void* add; const void* mul;
    int B, H, W, Cin, ldx, Ho, Wo, Cout, ldo, kh, kw, stride, pad_t, pad_l, dil, act, ld2;
};

// depthwise geometry: the 8-channel groups of an image are cut into nchunk chunks of cw groups (cw divides C / 8, cw <= 64), the pixels into
// nslab slabs; a block = (slab, chunk, image) of pl = 256 / cw pixel lanes x cw channel lanes
struct DwGeom { int cw, pl, nchunk, nslab, slab; };
DwGeom dw_geom(int B, int HW, int C) {
    DwGeom g;
    const int c8n = C / 8;
    g.nchunk = (c8n + 63) / 64;
    while (c8n % g.nchunk) ++g.nchunk;
    g.cw = c8n / g.nchunk;
    g.pl = 256 / g.cw;
    (void)B;                                                        // the slab cut decides the order of the SE sums: it must not depend on the batch
    long long want = 1024 / (long long)g.nchunk;                    // >= 1024 blocks per image: 256 CUs filled several waves deep at any batch
    want = want < 1 ? 1 : (want > 256 ? 256 : want);
    g.slab = (int)((HW + want - 1) / want);
    g.slab = ((g.slab + g.pl - 1) / g.pl) * g.pl;
    g.nslab = (HW + g.slab - 1) / g.slab;
    return g;
}

bool dw_strip(int kh, int kw, int stride, int dil) { return dil == 1 && kh == kw && (kh == 3 || kh == 5) && (stride == 1 || stride == 2); }

// depthwise: thread = (pixel lane, 8 consecutive channels), walking the slab's pixels pl apart; weights [kh][kw][C] fp32.  With `sums` the block
// also adds up its slab per channel, over the ROUNDED outputs, in a fixed order: sums[b][slab][c] -- the squeeze of squeeze-and-excite
template <class Tag>
__global__ __launch_bounds__(256) void k_seg_dwconv(const ConvD p, const int cw, const int pl, const int slab, float* __restrict__ sums) {
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    __shared__ float red[2048];
    const int c8l = threadIdx.x % cw, lp = threadIdx.x / cw;
    const int b = blockIdx.z, c0 = (blockIdx.y * cw + c8l) * 8;
    const int HW = p.Ho * p.Wo;
    const int q0 = blockIdx.x * slab, q1 = q0 + slab < HW ? q0 + slab : HW;
    float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (lp < pl) {
        float bias[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = p.bias ? p.bias[c0 + e] : 0.f;
        const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * p.H * p.W * p.ldx + c0;
        T* ob = reinterpret_cast<T*>(p.out) + (size_t)b * HW * p.ldo + c0;
        for (int q = q0 + lp; q < q1; q += pl) {
            const int yo = q / p.Wo, xo = q - yo * p.Wo;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = bias[e];
            for (int i = 0; i < p.kh; ++i) {
                const int yi = yo * p.stride - p.pad_t + i * p.dil;
                if (yi < 0 || yi >= p.H) continue;
                for (int j = 0; j < p.kw; ++j) {
                    const int xi = xo * p.stride - p.pad_l + j * p.dil;
                    if (xi < 0 || xi >= p.W) continue;
                    float v[8];
                    load8f<Tag>(xb + ((size_t)yi * p.W + xi) * p.ldx, v);
                    const float* wt = p.w + (size_t)(i * p.kw + j) * p.Cin + c0;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt), w1 = *reinterpret_cast<const f32x4*>(wt + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[e] = __builtin_fmaf(v[e], w0[e], acc[e]); acc[4 + e] = __builtin_fmaf(v[4 + e], w1[e], acc[4 + e]); }
                }
            }
            V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = Tag::from_f32(seg_act(acc[e], p.act)); tot[e] += Tag::to_f32(o[e]); }
            *reinterpret_cast<V8*>(ob + (size_t)q * p.ldo) = o;
        }
    }
    if (sums == nullptr) return;
    const int cwe = cw * 8;
    if (lp < pl) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[lp * cwe + c8l * 8 + e] = tot[e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < cwe; t += 256) {
        float a = 0.f;
        for (int k = 0; k < pl; ++k) a += red[k * cwe + t];
        sums[((size_t)b * gridDim.x + blockIdx.x) * p.Cin + blockIdx.y * cwe + t] = a;
    }
}

// the same for the encoder's four (kernel, stride) pairs, four outputs along x per thread: a row segment of (3 S + K) input vectors serves the
// four outputs and every tap's weights are fetched once per strip -- 2.25 loads per output instead of 6.75 (K = 3, S = 1).  Same arithmetic
// per output as k_seg_dwconv (taps in the same order, zero padding skipped the same way): bit-identical outputs.
template <class Tag, int K, int S>
__global__ __launch_bounds__(256, 4) void k_seg_dwconv_strip(const ConvD p, const int cw, const int pl, const int slab, float* __restrict__ sums) {
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    constexpr int XS = 4, NJ = (XS - 1) * S + K;
    __shared__ float red[2048];
    const int c8l = threadIdx.x % cw, lp = threadIdx.x / cw;
    const int b = blockIdx.z, c0 = (blockIdx.y * cw + c8l) * 8;
    const int spr = (p.Wo + XS - 1) / XS, nstrip = p.Ho * spr;            // strips per row / per image; `slab` counts strips here
    const int q0 = blockIdx.x * slab, q1 = q0 + slab < nstrip ? q0 + slab : nstrip;
    float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (lp < pl) {
        float bias[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = p.bias ? p.bias[c0 + e] : 0.f;
        const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * p.H * p.W * p.ldx + c0;
        T* ob = reinterpret_cast<T*>(p.out) + (size_t)b * p.Ho * p.Wo * p.ldo + c0;
        for (int q = q0 + lp; q < q1; q += pl) {
            const int yo = q / spr, xo0 = (q - yo * spr) * XS;
            float acc[XS][8];
#pragma unroll
            for (int u = 0; u < XS; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[u][e] = bias[e];
            const int xi0 = xo0 * S - p.pad_l;
#pragma unroll 1                                          // one row segment in registers at a time: <= 128 VGPRs, four waves per SIMD
            for (int i = 0; i < K; ++i) {
                const int yi = yo * S - p.pad_t + i;
                if (yi < 0 || yi >= p.H) continue;
                V8 seg[NJ];
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int xi = xi0 + jj;
                    if (xi >= 0 && xi < p.W) seg[jj] = *reinterpret_cast<const V8*>(xb + ((size_t)yi * p.W + xi) * p.ldx);
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) seg[jj][e] = (T)0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float* wt = p.w + (size_t)(i * K + j) * p.Cin + c0;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt), w1 = *reinterpret_cast<const f32x4*>(wt + 4);
#pragma unroll
                    for (int u = 0; u < XS; ++u) {
                        const int xi = xi0 + u * S + j;
                        if (xi < 0 || xi >= p.W) continue;          // a padded tap: skipped, as the one-output kernel skips it
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[u][e] = __builtin_fmaf(Tag::to_f32(seg[u * S + j][e]), w0[e], acc[u][e]);
                            acc[u][4 + e] = __builtin_fmaf(Tag::to_f32(seg[u * S + j][4 + e]), w1[e], acc[u][4 + e]);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < XS; ++u) {
                if (xo0 + u >= p.Wo) continue;
                V8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = Tag::from_f32(seg_act(acc[u][e], p.act)); tot[e] += Tag::to_f32(o[e]); }
                *reinterpret_cast<V8*>(ob + ((size_t)yo * p.Wo + xo0 + u) * p.ldo) = o;
            }
        }
    }
    if (sums == nullptr) return;
    const int cwe = cw * 8;
    if (lp < pl) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[lp * cwe + c8l * 8 + e] = tot[e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < cwe; t += 256) {
        float a = 0.f;
        for (int k = 0; k < pl; ++k) a += red[k * cwe + t];
        sums[((size_t)b * gridDim.x + blockIdx.x) * p.Cin + blockIdx.y * cwe + t] = a;
    }
}

// small dense convolution: a wave = 64 consecutive output pixels x CO output channels (weights are wave-uniform: [Cout][kh][kw][Cin] fp32);
// out = act(conv + bias) [* mul] [+ add], 16-bit output (or fp32 when OUT32)
template <class Tag, int CO, bool OUT32>
__global__ __launch_bounds__(256) void k_seg_conv(const ConvD p) {
    typedef typename Tag::T T;
    const long long npix = (long long)p.B * p.Ho * p.Wo;
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * CO;
    if (pix >= npix) return;
    const int xo = (int)(pix % p.Wo), yo = (int)((pix / p.Wo) % p.Ho), b = (int)(pix / ((long long)p.Wo * p.Ho));
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = (p.bias && co0 + o < p.Cout) ? p.bias[co0 + o] : 0.f;
    const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * p.H * p.W * p.ldx;
    const int taps = p.kh * p.kw;
    for (int i = 0; i < p.kh; ++i) {
        const int yi = yo * p.stride - p.pad_t + i * p.dil;
        if (yi < 0 || yi >= p.H) continue;
        for (int j = 0; j < p.kw; ++j) {
            const int xi = xo * p.stride - p.pad_l + j * p.dil;
            if (xi < 0 || xi >= p.W) continue;
            const T* xp = xb + ((size_t)yi * p.W + xi) * p.ldx;
            const float* wt = p.w + ((size_t)co0 * taps + (i * p.kw + j)) * p.Cin;
            int c = 0;
            if ((p.Cin & 7) == 0 && (p.ldx & 7) == 0) {
                for (; c < p.Cin; c += 8) {
                    float v[8];
                    load8f<Tag>(xp + c, v);
#pragma unroll
                    for (int o = 0; o < CO; ++o) {
                        if (co0 + o < p.Cout) {
                            const float* wr = wt + (size_t)o * taps * p.Cin + c;
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[o] = __builtin_fmaf(v[e], wr[e], acc[o]);
                        }
                    }
                }
            }
            for (; c < p.Cin; ++c) {
                const float v = Tag::to_f32(xp[c]);
#pragma unroll
                for (int o = 0; o < CO; ++o)
                    if (co0 + o < p.Cout) acc[o] = __builtin_fmaf(v, wt[(size_t)o * taps * p.Cin + c], acc[o]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        if (co0 + o >= p.Cout) break;
        float v = seg_act(acc[o], p.act);
        const size_t o2 = (size_t)pix * p.ld2 + co0 + o;        // mul / add are fp32 tensors when the output is
        if (p.mul) v *= OUT32 ? reinterpret_cast<const float*>(p.mul)[o2] : Tag::to_f32(reinterpret_cast<const T*>(p.mul)[o2]);
        if (p.add) v += OUT32 ? reinterpret_cast<const float*>(p.add)[o2] : Tag::to_f32(reinterpret_cast<const T*>(p.add)[o2]);
        if (OUT32) reinterpret_cast<float*>(p.out)[(size_t)pix * p.ldo + co0 + o] = v;
        else reinterpret_cast<T*>(p.out)[(size_t)pix * p.ldo + co0 + o] = Tag::from_f32(v);
    }
}

// Small dense convolutions on the matrix cores: the 1 x 1 expand / project layers of EfficientNet (K, N from 32 to 3840) and the decoder's
// 1 x k, k x 1 and dilated 3 x 3 layers (stride 1, "same" size), all HBM- or latency-bound skinny GEMMs:
//   out[m][n] = act(sum_{tap, c} (x[pixel(m) + tap][c] * gate[b][c]) W[n][tap][c] + bias[n]) (+ residual[m][n]).
// A wave tile is 64 pixels x 64 output channels (4 x 4 fragments of v_mfma_f32_16x16x32).  W is the MFMA's row operand and the pixels its
// column operand, so that a lane ends up with FOUR CONSECUTIVE CHANNELS of one pixel (8-byte stores, no LDS transpose), and both operands
// are plain 16-byte global loads in the layout the instruction wants (lane = row l % 16, k = 8 (l / 16) .. + 7): no LDS staging.  The
// squeeze-and-excite gate multiplies the pixel fragments in registers (rounded to the storage type, as the reference's x * gate is): the SE
// scaling costs no pass of its own.  The next K step's fragments are loaded before the current one's MFMAs (register double buffer).
// Block = 4 waves on ONE image (the gate is per image):
//   SPLITK = false: 256 pixels x 64 channels, a wave per 64 pixels;
//   SPLITK = true : 64 pixels x 64 channels, the waves take every fourth K step and the four partial tiles meet through LDS in wave order
//                   (deterministic) -- for the deep layers, where an image has few pixels and K is in the thousands.
// The choice depends on (K, pixels per image) only, never on the batch.  Grid (N tiles, pixel tiles x images): the channel tiles of one
// pixel block are neighbours in launch order and find its rows in L2.
__device__ __forceinline__ void lane_swap16(float& x, float& y) {      // x of lanes 16..31 / 48..63 <-> y of lanes 0..15 / 32..47
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void lane_swap32(float& x, float& y) {      // x of lanes 32..63 <-> y of lanes 0..31
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}

struct PwD {
    const void* x; const void* w; const float* bias; const float* gate; const void* res; void* out;
    int B, HW, K, N, ldx, ldw, ldo, ldr, act;
    int H, W, Cin, kh, kw, dil, pad_t, pad_l;             // taps (kh * kw > 1): K = kh * kw * Cin, Cin % 32 == 0
};

template <class Tag, bool SPLITK, bool TAPS>
__global__ __launch_bounds__(256, SPLITK ? 2 : 4) void k_seg_mconv(const PwD p) {
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    typedef T T4 __attribute__((ext_vector_type(4)));
    extern __shared__ f32x4 red[];                        // SPLITK: [4 waves][16 fragments][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kg = lane >> 4;
    constexpr int PXB = SPLITK ? 64 : 256;
    const int tiles = (p.HW + PXB - 1) / PXB;
    const int b = blockIdx.y / tiles, px0 = (blockIdx.y - b * tiles) * PXB + (SPLITK ? 0 : wave * 64);
    if (!SPLITK && px0 >= p.HW) return;
    const int n0 = blockIdx.x * 64;
    const int nfv = (p.N - n0 + 15) / 16 < 4 ? (p.N - n0 + 15) / 16 : 4;
    const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * p.HW * p.ldx + kg * 8;
    int pxl[4], py[4], pxx[4];
    const T* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int px = px0 + i * 16 + r16;
        px = px < p.HW ? px : p.HW - 1;
        pxl[i] = px;
        if (TAPS) { py[i] = px / p.W; pxx[i] = px - py[i] * p.W; }
        int n = n0 + i * 16 + r16;
        n = n < p.N ? n : p.N - 1;
        wp[i] = reinterpret_cast<const T*>(p.w) + (size_t)n * p.ldw + kg * 8;
    }
    const float* gp = p.gate ? p.gate + (size_t)b * p.K + kg * 8 : nullptr;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int KSTEP = SPLITK ? 128 : 32;
    auto load = [&](int k0, V8 (&xf)[4], V8 (&wf)[4]) {
        const bool kv = k0 + kg * 8 + 8 <= p.K;            // K % 8 == 0: a lane's eight k are in or out together
        int dy = 0, dx = 0, kin = k0;
        if (TAPS) {
            const int t = k0 / p.Cin;
            kin = k0 - t * p.Cin;
            const int ty = t / p.kw;
            dy = ty * p.dil - p.pad_t;
            dx = (t - ty * p.kw) * p.dil - p.pad_l;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { xf[i][e] = (T)0.f; wf[i][e] = (T)0.f; }
            if (kv) {
                if (TAPS) {
                    const int yy = py[i] + dy, xx = pxx[i] + dx;
                    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) xf[i] = *reinterpret_cast<const V8*>(xb + ((size_t)yy * p.W + xx) * p.ldx + kin);
                } else {
                    xf[i] = *reinterpret_cast<const V8*>(xb + (size_t)pxl[i] * p.ldx + k0);
                }
                if (i < nfv) wf[i] = *reinterpret_cast<const V8*>(wp[i] + k0);
            }
        }
        if (gp && kv) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp + k0), g1 = *reinterpret_cast<const f32x4*>(gp + k0 + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xf[i][e] = Tag::from_f32(Tag::to_f32(xf[i][e]) * g0[e]);
                    xf[i][4 + e] = Tag::from_f32(Tag::to_f32(xf[i][4 + e]) * g1[e]);
                }
        }
    };
    auto mma = [&](const V8 (&xf)[4], const V8 (&wf)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nfv) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = Tag::mfma16(wf[j], xf[i], acc[i][j]);
            }
        }
    };
    if constexpr (SPLITK) {                               // long K, few waves: the next step's fragments fly while this one's MFMAs run
        V8 xa[4], wa[4], xc[4], wc[4];
        int k0 = wave * 32;
        if (k0 < p.K) load(k0, xa, wa);
        while (k0 < p.K) {
            if (k0 + KSTEP < p.K) load(k0 + KSTEP, xc, wc);
            mma(xa, wa);
            k0 += KSTEP;
            if (k0 >= p.K) break;
            if (k0 + KSTEP < p.K) load(k0 + KSTEP, xa, wa);
            mma(xc, wc);
            k0 += KSTEP;
        }
    } else {                                              // streaming shapes: <= 128 registers, four waves per SIMD hide the latency instead
        V8 xa[4], wa[4];
        for (int k0 = 0; k0 < p.K; k0 += KSTEP) {
            load(k0, xa, wa);
            mma(xa, wa);
        }
    }
    if (SPLITK) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(wave * 16 + i * 4 + j) * 64 + lane] = acc[i][j];
        __syncthreads();
        // wave w finishes pixel fragment w of the tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = red[(0 * 16 + wave * 4 + j) * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) v += red[(ww * 16 + wave * 4 + j) * 64 + lane];
            acc[0][j] = v;
        }
    }
    // a lane holds, per pixel fragment i, channels n0 + 16 j + 4 kg .. + 3 of pixel 16 i + r16 (j = 0 .. 3).  Bias and activation go on in this
    // layout; then the four lanes of a pixel (kg = 0 .. 3) transpose their 4 x 4 block of f32x4 with v_permlane16_swap / v_permlane32_swap so
    // that lane kg owns the SIXTEEN consecutive channels of fragment j = kg: residual and output move as 32 contiguous bytes per lane, 128 per
    // pixel -- whole cache lines instead of 32-byte pieces
    const bool wide = (p.ldo & 7) == 0 && (!p.res || (p.ldr & 7) == 0);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        if (SPLITK && ii > 0) break;
        const int i = SPLITK ? wave : ii;
        const int px = px0 + i * 16 + r16;
        float v[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            f32x4 t = acc[ii][jj];
            const int n = n0 + jj * 16 + kg * 4;
            if (p.bias && jj < nfv && n < p.N) { const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n); t += b4; }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[jj][e] = seg_act(t[e], p.act);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lane_swap16(v[0][e], v[1][e]); lane_swap16(v[2][e], v[3][e]);
            lane_swap32(v[0][e], v[2][e]); lane_swap32(v[1][e], v[3][e]);
        }
        if (px >= p.HW || kg >= nfv) continue;
        const size_t row = (size_t)b * p.HW + px;
        const int nb = n0 + kg * 16;                       // v[q][e]: channel nb + 4 q + e
        T* op = reinterpret_cast<T*>(p.out) + row * p.ldo + nb;
        const T* rp = p.res ? reinterpret_cast<const T*>(p.res) + row * p.ldr + nb : nullptr;
        if (wide && nb + 16 <= p.N) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                V8 o;
                if (rp) {
                    const V8 r8 = *reinterpret_cast<const V8*>(rp + 8 * h);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = Tag::from_f32(v[2 * h + (e >> 2)][e & 3] + Tag::to_f32(r8[e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = Tag::from_f32(v[2 * h + (e >> 2)][e & 3]);
                }
                *reinterpret_cast<V8*>(op + 8 * h) = o;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (nb + 4 * q >= p.N) continue;           // N % 4 == 0
                T4 o;
                if (rp) {
                    const T4 r4 = *reinterpret_cast<const T4*>(rp + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = Tag::from_f32(v[q][e] + Tag::to_f32(r4[e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = Tag::from_f32(v[q][e]);
                }
                *reinterpret_cast<T4*>(op + 4 * q) = o;
            }
        }
    }
}

template <class Tag>
int mconv_run(const PwD& p, hipStream_t s) {
    const bool taps = p.kh * p.kw > 1;
    const bool splitk = p.K >= 256 && p.HW <= 6400;       // a function of the layer, never of the batch
    const dim3 grid(mve_cdiv(p.N, 64), (unsigned)(mve_cdiv(p.HW, splitk ? 64 : 256) * p.B));
    const size_t lds = splitk ? 4 * 16 * 64 * sizeof(f32x4) : 0;
    if (splitk) {
        static bool configured[64] = {};
        int dev = 0;
        MVE_HIP(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !configured[dev]) {
            MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg_mconv<Tag, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg_mconv<Tag, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured[dev] = true;
        }
        if (taps) k_seg_mconv<Tag, true, true><<<grid, 256, lds, s>>>(p); else k_seg_mconv<Tag, true, false><<<grid, 256, lds, s>>>(p);
    } else {
        if (taps) k_seg_mconv<Tag, false, true><<<grid, 256, 0, s>>>(p); else k_seg_mconv<Tag, false, false><<<grid, 256, 0, s>>>(p);
    }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

template <class Tag>
__global__ __launch_bounds__(256) void k_seg_act(void* x, size_t n8, int act) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    V8* p = reinterpret_cast<V8*>(x) + i;
    V8 v = *p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = Tag::from_f32(seg_act(Tag::to_f32(v[e]), act));
    *p = v;
}

// [B, HW, C] -> mean over HW: grid (C / 8 chunks, B), 256 threads stride over the pixels, fixed-order tree (deterministic)
template <class Tag>
__global__ __launch_bounds__(256) void k_seg_channel_mean(const void* x, int HW, int C, float* out) {
    typedef typename Tag::T T;
    __shared__ float red[256][8];
    const int c0 = blockIdx.x * 8, b = blockIdx.y;
    const T* xb = reinterpret_cast<const T*>(x) + (size_t)b * HW * C + c0;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = threadIdx.x; p < HW; p += 256) {
        float v[8];
        load8f<Tag>(xb + (size_t)p * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[threadIdx.x + d][e];
        }
        __syncthreads();
    }
    if (threadIdx.x < 8) out[(size_t)b * C + c0 + threadIdx.x] = red[0][threadIdx.x] / (float)HW;
}

// squeeze-and-excite gate in two launches, a wave per output so that every weight row is read once, coalesced:
//   hidden[b][s] = swish(b1[s] + W1[s][:] . pooled[b])                                                 (grid (S / 4, B))
//   gate[b][c]   = sigmoid(b2[c] + W2[c][:] . hidden[b])                                               (grid (C / 4, B))
// the lane-strided partial sums meet in a fixed xor butterfly: deterministic
__device__ __forceinline__ float wave_sum(float a) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
    return a;
}

// pooled[b][c] = scale * sum_k sums[b][k][c]: block = (64 channels, image); wave w adds up the slabs k = w, w + 4, ... (lanes = channels:
// coalesced), the four partial sums meet in wave order -- fixed order, deterministic
__global__ __launch_bounds__(256) void k_seg_pool_finalize(const float* __restrict__ sums, int nslab, float scale, int C, float* __restrict__ pooled) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float a = 0.f;
    if (c < C)
        for (int k = w; k < nslab; k += 4) a += sums[((size_t)b * nslab + k) * C + c];
    part[w][threadIdx.x & 63] = a;
    __syncthreads();
    if (w == 0 && c < C) pooled[(size_t)b * C + c] = (((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x]) * scale;
}

__global__ __launch_bounds__(256) void k_seg_se_hidden(const float* __restrict__ pooled, int C, int S, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, float* __restrict__ hidden) {
    const int b = blockIdx.y;
    const float* pv = pooled + (size_t)b * C;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= S) return;
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a = __builtin_fmaf(w1[(size_t)s * C + c], pv[c], a);
    a = wave_sum(a) + b1[s];
    if (lane == 0) hidden[(size_t)b * S + s] = a / (1.0f + __expf(-a));
}

__global__ __launch_bounds__(256) void k_seg_se_out(const float* __restrict__ hidden, int C, int S, const float* __restrict__ w2, const float* __restrict__ b2,
                                                    float* __restrict__ gate) {
    const int b = blockIdx.y, c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float a = 0.f;
    for (int s = lane; s < S; s += 64) a = __builtin_fmaf(w2[(size_t)c * S + s], hidden[(size_t)b * S + s], a);
    a = wave_sum(a) + b2[c];
    if (lane == 0) gate[(size_t)b * C + c] = 1.0f / (1.0f + __expf(-a));
}

// x[b, p, c] = x * A[b, c] (+ S[b, c])
template <class Tag>
__global__ __launch_bounds__(256) void k_seg_scale(void* x, const void* src, int HW, int C, const float* A, const float* S, size_t n8) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c8n = C / 8;
    const int c0 = (int)(i % c8n) * 8;
    const int b = (int)(i / ((size_t)c8n * HW));
    V8 v = reinterpret_cast<const V8*>(src)[i];
    const float* a = A + (size_t)b * C + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = Tag::to_f32(v[e]) * a[e];
        if (S) f += S[(size_t)b * C + c0 + e];
        v[e] = Tag::from_f32(f);
    }
    reinterpret_cast<V8*>(x)[i] = v;
}

// bilinear sample position of output index o (PyTorch upsample_bilinear2d / torchvision resize without antialias)
__device__ __forceinline__ void bil_pos(int o, int n_in, int n_out, int align, int& i0, int& i1, float& f) {
    float src;
    if (align) src = n_out > 1 ? o * (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    else { src = (o + 0.5f) * ((float)n_in / (float)n_out) - 0.5f; src = src < 0.f ? 0.f : src; }
    i0 = (int)src;
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    f = src - (float)i0;
}

// mode bit 0: input is NCHW fp32 (else NHWC of IN type); normalisation (x - mean[c]) / std[c] when mean != null
template <class TagI, class TagO, bool IN32, bool OUT32>
__global__ __launch_bounds__(256) void k_seg_resize(const void* x, int B, int H, int W, int C, void* out, int Ho, int Wo, int align, int in_nchw,
                                                    const float* mean, const float* stdv) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * Ho * Wo * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), b = (int)(pix / ((long long)Wo * Ho));
    int y0, y1, x0, x1;
    float fy, fx;
    bil_pos(yo, H, Ho, align, y0, y1, fy);
    bil_pos(xo, W, Wo, align, x0, x1, fx);
    auto at = [&](int yy, int xx) -> float {
        if (in_nchw) return reinterpret_cast<const float*>(x)[(((size_t)b * C + c) * H + yy) * W + xx];
        const size_t o = (((size_t)b * H + yy) * W + xx) * C + c;
        if (IN32) return reinterpret_cast<const float*>(x)[o];
        return TagI::to_f32(reinterpret_cast<const typename TagI::T*>(x)[o]);
    };
    const float top = at(y0, x0) + (at(y0, x1) - at(y0, x0)) * fx, bot = at(y1, x0) + (at(y1, x1) - at(y1, x0)) * fx;
    float v = top + (bot - top) * fy;
    if (mean) v = (v - mean[c]) / stdv[c];
    if (OUT32) reinterpret_cast<float*>(out)[idx] = v;
    else reinterpret_cast<typename TagO::T*>(out)[idx] = TagO::from_f32(v);
}

// union attention, channel tracer.  One block per image, C <= 256 threads active.  pooled [B][C]; bn: xn = pooled * ns + nb; att out [B][C],
// mask out [B][C] (att where att > its 10 % quantile, else 0)
__global__ __launch_bounds__(256) void k_seg_uam_channel(const float* pooled, int C, const float* ns, const float* nb, const float* wq, const float* wk,
                                                         const float* wv, const float* wfc, float ratio, const float* bs, const float* bt,
                                                         float* att, float* A, float* S) {
    __shared__ float xn[256], q[256], k[256], v[256], o[256], a[256];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < C) xn[t] = pooled[(size_t)b * C + t] * ns[t] + nb[t];
    __syncthreads();
    if (t < C) {
        float aq = 0.f, ak = 0.f, av = 0.f;
        for (int c = 0; c < C; ++c) {
            aq = __builtin_fmaf(wq[(size_t)t * C + c], xn[c], aq);
            ak = __builtin_fmaf(wk[(size_t)t * C + c], xn[c], ak);
            av = __builtin_fmaf(wv[(size_t)t * C + c], xn[c], av);
        }
        q[t] = aq; k[t] = ak; v[t] = av;
    }
    __syncthreads();
    if (t < C) {            // row t of softmax(q k^T) v: scores q[t] * k[j]
        float m = -INFINITY;
        for (int j = 0; j < C; ++j) m = fmaxf(m, q[t] * k[j]);
        float s = 0.f, acc = 0.f;
        for (int j = 0; j < C; ++j) { const float e = __expf(q[t] * k[j] - m); s += e; acc = __builtin_fmaf(e, v[j], acc); }
        o[t] = acc / s;
    }
    __syncthreads();
    if (t < C) {
        float f = 0.f;
        for (int c = 0; c < C; ++c) f = __builtin_fmaf(wfc[(size_t)t * C + c], o[c], f);
        a[t] = 1.0f / (1.0f + __expf(-f));
    }
    __syncthreads();
    // torch.quantile(mask, ratio) with linear interpolation: position ratio * (C - 1) in the sorted values
    __shared__ float sorted[256];
    if (t < C) {
        int rank = 0;
        for (int j = 0; j < C; ++j) rank += (a[j] < a[t]) || (a[j] == a[t] && j < t);
        sorted[rank] = a[t];
    }
    __syncthreads();
    if (t < C) {
        const float pos = ratio * (float)(C - 1);
        const int lo = (int)floorf(pos), hi = lo + 1 < C ? lo + 1 : C - 1;
        const float thr = sorted[lo] + (sorted[hi] - sorted[lo]) * (pos - (float)lo);
        const float m = a[t] <= thr ? 0.f : a[t];
        att[(size_t)b * C + t] = a[t];
        // x_drop = (BatchNorm(x * att + x)) * mask = x * ((1 + att) * bs * m) + bt * m   (att_modules.py:163-174)
        A[(size_t)b * C + t] = (1.0f + a[t]) * bs[t] * m;
        S[(size_t)b * C + t] = bt[t] * m;
    }
}

// spatial union attention: qkv [B][H*W][3] fp32 -> out [B][H*W] = softmax_rows(q k^T) v + v, q / k / v seen as H x W matrices.
// One block per (image, row i): scores s_j = sum_w q[i][w] k[j][w]
__global__ __launch_bounds__(256) void k_seg_uam_spatial(const float* qkv, int H, int W, float* out) {
    extern __shared__ float sh[];          // q row [W], scores [H]
    float* qrow = sh;
    float* sc = sh + W;
    const int b = blockIdx.y, i = blockIdx.x, t = threadIdx.x;
    const float* base = qkv + (size_t)b * H * W * 3;
    for (int w = t; w < W; w += 256) qrow[w] = base[((size_t)i * W + w) * 3];
    __syncthreads();
    for (int j = t; j < H; j += 256) {
        float s = 0.f;
        for (int w = 0; w < W; ++w) s = __builtin_fmaf(qrow[w], base[((size_t)j * W + w) * 3 + 1], s);
        sc[j] = s;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = 0; j < H; ++j) m = fmaxf(m, sc[j]);
    float den = 0.f;
    for (int j = 0; j < H; ++j) den += __expf(sc[j] - m);
    for (int w = t; w < W; w += 256) {
        float acc = 0.f;
        for (int j = 0; j < H; ++j) acc = __builtin_fmaf(__expf(sc[j] - m), base[((size_t)j * W + w) * 3 + 2], acc);
        out[(size_t)b * H * W + (size_t)i * W + w] = acc / den + base[((size_t)i * W + w) * 3 + 2];
    }
}

// x = enc * (sigmoid(d) + edge), edge = (1 - sigmoid(d)) where that is <= 0.93 (else 0)
template <class Tag>
__global__ __launch_bounds__(256) void k_seg_object_mix(const float* d, const void* enc, void* out, int C, size_t n8) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const size_t pix = i / (C / 8);
    const float ob = 1.0f / (1.0f + __expf(-d[pix]));
    float bg = 1.0f - ob;
    if (bg > 0.93f) bg = 0.f;
    const float f = ob + bg;
    V8 v = reinterpret_cast<const V8*>(enc)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = Tag::from_f32(Tag::to_f32(v[e]) * f);
    reinterpret_cast<V8*>(out)[i] = v;
}

// out = x * y (* z)
template <class Tag>
__global__ __launch_bounds__(256) void k_seg_mul(const void* x, const void* y, const void* z, void* out, size_t n8) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const V8 a = reinterpret_cast<const V8*>(x)[i], b = reinterpret_cast<const V8*>(y)[i];
    V8 c = a;
    if (z) c = reinterpret_cast<const V8*>(z)[i];
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = Tag::to_f32(a[e]) * Tag::to_f32(b[e]);
        if (z) v *= Tag::to_f32(c[e]);
        o[e] = Tag::from_f32(v);
    }
    reinterpret_cast<V8*>(out)[i] = o;
}

__device__ __forceinline__ float bil_sample(const float* m, int H, int W, int Ho, int Wo, int yo, int xo) {
    int y0, y1, x0, x1;
    float fy, fx;
    bil_pos(yo, H, Ho, 0, y0, y1, fy);
    bil_pos(xo, W, Wo, 0, x0, x1, fx);
    const float top = m[y0 * W + x0] + (m[y0 * W + x1] - m[y0 * W + x0]) * fx, bot = m[y1 * W + x0] + (m[y1 * W + x1] - m[y1 * W + x0]) * fx;
    return top + (bot - top) * fy;
}

// out[b][y][x] = sigmoid((up8(d0) + up8(d1) + up4(d2)) / 3): d0, d1 [B][S/8][S/8], d2 [B][S/4][S/4]
__global__ __launch_bounds__(256) void k_seg_fuse(const float* d0, const float* d1, const float* d2, int B, int Hs, int Ws, float* out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * Hs * Ws) return;
    const int xo = (int)(idx % Ws), yo = (int)((idx / Ws) % Hs), b = (int)(idx / ((long long)Ws * Hs));
    const int h8 = Hs / 8, w8 = Ws / 8, h4 = Hs / 4, w4 = Ws / 4;
    const float m0 = bil_sample(d0 + (size_t)b * h8 * w8, h8, w8, Hs, Ws, yo, xo);
    const float m1 = bil_sample(d1 + (size_t)b * h8 * w8, h8, w8, Hs, Ws, yo, xo);
    const float m2 = bil_sample(d2 + (size_t)b * h4 * w4, h4, w4, Hs, Ws, yo, xo);
    // the reference adds in the order (ds_map2 + ds_map1 + ds_map0) / 3
    out[idx] = 1.0f / (1.0f + __expf(-((m2 + m1 + m0) / 3.0f)));
}

// 3x3 (2 r + 1) min-pool of m [B][Hs][Ws] (the reference's -max_pool(-m) with implicit -inf padding: border windows shrink), then bilinear
// resize to [Ho][Wo]; fail[b] is cleared when some output pixel is <= 0.2
__global__ __launch_bounds__(256) void k_seg_erode(const float* m, int B, int Hs, int Ws, int r, float* out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * Hs * Ws) return;
    const int xo = (int)(idx % Ws), yo = (int)((idx / Ws) % Hs), b = (int)(idx / ((long long)Ws * Hs));
    float v = INFINITY;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int y = yo + dy, x = xo + dx;
            if (y >= 0 && y < Hs && x >= 0 && x < Ws) v = fminf(v, m[((size_t)b * Hs + y) * Ws + x]);
        }
    out[idx] = v;
}
template <class Tag>
__global__ __launch_bounds__(256) void k_seg_post_resize(const float* m, int B, int Hs, int Ws, int Ho, int Wo, float* out, int* not_failed, int out_f32) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * Ho * Wo) return;
    const int xo = (int)(idx % Wo), yo = (int)((idx / Wo) % Ho), b = (int)(idx / ((long long)Wo * Ho));
    float v = bil_sample(m + (size_t)b * Hs * Ws, Hs, Ws, Ho, Wo, yo, xo);
    if (!out_f32) v = Tag::to_f32(Tag::from_f32(v));      // the 16-bit module compares the rounded mask
    out[idx] = v;
    if (!(v > 0.2f)) not_failed[b] = 1;          // benign race: every writer stores 1
}
template <class Tag>
__global__ __launch_bounds__(256) void k_seg_post_apply(const float* m, int B, int HW, const int* not_failed, void* out, int out_f32) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * HW) return;
    const int b = (int)(idx / HW);
    float v = m[idx];
    if (!not_failed[b] && v < 0.8f) v = 0.f;
    if (out_f32) reinterpret_cast<float*>(out)[idx] = v;
    else reinterpret_cast<typename Tag::T*>(out)[idx] = Tag::from_f32(v);
}

template <class Tag>
int conv2d_run(const ConvD& p, int depthwise, int out_f32, hipStream_t s, float* sums = nullptr) {
    if (depthwise) {
        const bool strip = dw_strip(p.kh, p.kw, p.stride, p.dil);
        const DwGeom g = dw_geom(p.B, strip ? p.Ho * ((p.Wo + 3) / 4) : p.Ho * p.Wo, p.Cin);
        const dim3 grid(g.nslab, g.nchunk, p.B);
        if (!strip) k_seg_dwconv<Tag><<<grid, 256, 0, s>>>(p, g.cw, g.pl, g.slab, sums);
        else if (p.kh == 3 && p.stride == 1) k_seg_dwconv_strip<Tag, 3, 1><<<grid, 256, 0, s>>>(p, g.cw, g.pl, g.slab, sums);
        else if (p.kh == 3) k_seg_dwconv_strip<Tag, 3, 2><<<grid, 256, 0, s>>>(p, g.cw, g.pl, g.slab, sums);
        else if (p.stride == 1) k_seg_dwconv_strip<Tag, 5, 1><<<grid, 256, 0, s>>>(p, g.cw, g.pl, g.slab, sums);
        else k_seg_dwconv_strip<Tag, 5, 2><<<grid, 256, 0, s>>>(p, g.cw, g.pl, g.slab, sums);
    } else {
        const long long npix = (long long)p.B * p.Ho * p.Wo;
        const int CO = p.Cout >= 8 ? 8 : 4;
        dim3 grid(mve_cdiv(npix, 256), mve_cdiv(p.Cout, CO));
        if (out_f32) {
            if (CO == 8) k_seg_conv<Tag, 8, true><<<grid, 256, 0, s>>>(p); else k_seg_conv<Tag, 4, true><<<grid, 256, 0, s>>>(p);
        } else {
            if (CO == 8) k_seg_conv<Tag, 8, false><<<grid, 256, 0, s>>>(p); else k_seg_conv<Tag, 4, false><<<grid, 256, 0, s>>>(p);
        }
    }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

template <class Tag>
int resize_run(const void* x, int B, int H, int W, int C, void* out, int Ho, int Wo, int align, int in_mode, int out_f32, const float* mean,
               const float* stdv, hipStream_t s) {
    const long long total = (long long)B * Ho * Wo * C;
    const unsigned grid = mve_cdiv(total, 256);
    const bool in32 = in_mode != 0;
    const int nchw = in_mode == 2;
    if (in32 && out_f32) k_seg_resize<Tag, Tag, true, true><<<grid, 256, 0, s>>>(x, B, H, W, C, out, Ho, Wo, align, nchw, mean, stdv);
    else if (in32) k_seg_resize<Tag, Tag, true, false><<<grid, 256, 0, s>>>(x, B, H, W, C, out, Ho, Wo, align, nchw, mean, stdv);
    else if (out_f32) k_seg_resize<Tag, Tag, false, true><<<grid, 256, 0, s>>>(x, B, H, W, C, out, Ho, Wo, align, nchw, mean, stdv);
    else k_seg_resize<Tag, Tag, false, false><<<grid, 256, 0, s>>>(x, B, H, W, C, out, Ho, Wo, align, nchw, mean, stdv);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // namespace

extern "C" {

int mve_seg_conv2d(int dtype, const void* x, int B, int H, int W, int Cin, int ldx, const float* w, const float* bias, void* out, int Ho, int Wo,
                   int Cout, int ldo, int kh, int kw, int stride, int pad_t, int pad_l, int dil, int depthwise, int act, const void* mul,
                   const void* add, int ld2, int out_f32, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(x && w && out && Cin > 0 && Cout > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0, MVE_ERR_ARG, "seg_conv2d: bad arguments");
    MVE_CHECK(!depthwise || (Cin == Cout && Cin % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && !mul && !add && !out_f32), MVE_ERR_ARG,
              "seg_conv2d: depthwise needs Cin == Cout, channel counts / strides that are multiples of 8 and no mul / add / fp32 output");
    ConvD p;
    p.x = x; p.w = w; p.bias = bias; p.out = out; p.add = add; p.mul = mul;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.ldx = ldx; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.ldo = ldo; p.kh = kh; p.kw = kw;
    p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l; p.dil = dil; p.act = act; p.ld2 = ld2;
    if (dtype == MVE_F16) return conv2d_run<F16Tag>(p, depthwise, out_f32, (hipStream_t)stream);
    if (dtype == MVE_BF16) return conv2d_run<BF16Tag>(p, depthwise, out_f32, (hipStream_t)stream);
    mve_set_error("seg_conv2d: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

int mve_seg_mconv(int dtype, const void* x, int B, int H, int W, int Cin, int ldx, const void* w, int ldw, int kh, int kw, int dil, int pad_t,
                  int pad_l, const float* bias, const float* gate, const void* residual, int ldr, void* out, int N, int ldo, int act, void* stream) {
    if (B == 0 || H == 0 || W == 0) return MVE_OK;
    const int taps = kh * kw;
    MVE_CHECK(x && w && out && Cin > 0 && N > 0 && kh > 0 && kw > 0 && dil > 0 && Cin % 8 == 0 && N % 4 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldo % 4 == 0 &&
                  (!residual || ldr % 4 == 0), MVE_ERR_ARG, "seg_mconv: Cin, ldx, ldw must be multiples of 8, N, ldo, ldr of 4");
    MVE_CHECK(taps == 1 || Cin % 32 == 0, MVE_ERR_ARG, "seg_mconv: a kernel with taps needs Cin %% 32 == 0 (got %d)", Cin);
    MVE_CHECK(taps == 1 || !gate, MVE_ERR_ARG, "seg_mconv: the input gate goes with 1 x 1 kernels only");
    MVE_CHECK(act >= 0 && act <= 4, MVE_ERR_ARG, "seg_mconv: act in 0..4");
    PwD p;
    p.x = x; p.w = w; p.bias = bias; p.gate = gate; p.res = residual; p.out = out;
    p.B = B; p.HW = H * W; p.K = taps * Cin; p.N = N; p.ldx = ldx; p.ldw = ldw; p.ldo = ldo; p.ldr = ldr; p.act = act;
    p.H = H; p.W = W; p.Cin = Cin; p.kh = kh; p.kw = kw; p.dil = dil; p.pad_t = pad_t; p.pad_l = pad_l;
    if (dtype == MVE_F16) return mconv_run<F16Tag>(p, (hipStream_t)stream);
    if (dtype == MVE_BF16) return mconv_run<BF16Tag>(p, (hipStream_t)stream);
    mve_set_error("seg_mconv: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

int mve_seg_act(int dtype, void* x, size_t n, int act, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(x && n % 8 == 0 && act >= 0 && act <= 4, MVE_ERR_ARG, "seg_act: n must be a multiple of 8, act in 0..4");
    const unsigned grid = mve_cdiv(n / 8, 256);
    if (dtype == MVE_F16) k_seg_act<F16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(x, n / 8, act);
    else if (dtype == MVE_BF16) k_seg_act<BF16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(x, n / 8, act);
    else { mve_set_error("seg_act: unsupported dtype %d", dtype); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_seg_channel_mean(int dtype, const void* x, int B, int HW, int C, float* out, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(x && out && C % 8 == 0 && HW > 0, MVE_ERR_ARG, "seg_channel_mean: C must be a multiple of 8");
    dim3 grid(C / 8, B);
    if (dtype == MVE_F16) k_seg_channel_mean<F16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(x, HW, C, out);
    else if (dtype == MVE_BF16) k_seg_channel_mean<BF16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(x, HW, C, out);
    else { mve_set_error("seg_channel_mean: unsupported dtype %d", dtype); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_seg_se_gate(const float* sums, int nslab, float scale, int B, int C, int S, const float* w1, const float* b1, const float* w2, const float* b2,
                    float* hidden, float* gate, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(sums && w1 && b1 && w2 && b2 && hidden && gate && nslab > 0 && C > 0 && C <= 8192 && S > 0, MVE_ERR_ARG, "seg_se_gate: bad arguments");
    const float* pooled = sums;
    if (nslab != 1 || scale != 1.0f) {                     // hidden: [B][S] followed by [B][C] for the pooled vector
        float* pl = hidden + (size_t)B * S;
        k_seg_pool_finalize<<<dim3(mve_cdiv(C, 64), B), 256, 0, (hipStream_t)stream>>>(sums, nslab, scale, C, pl);
        MVE_LAUNCH_CHECK();
        pooled = pl;
    }
    k_seg_se_hidden<<<dim3(mve_cdiv(S, 4), B), 256, 0, (hipStream_t)stream>>>(pooled, C, S, w1, b1, hidden);
    MVE_LAUNCH_CHECK();
    k_seg_se_out<<<dim3(mve_cdiv(C, 4), B), 256, 0, (hipStream_t)stream>>>(hidden, C, S, w2, b2, gate);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_seg_dwconv_slabs(int B, int Ho, int Wo, int C, int k, int stride) {
    if (B <= 0 || Ho <= 0 || Wo <= 0 || C < 8) return 0;
    return dw_geom(B, dw_strip(k, k, stride, 1) ? Ho * ((Wo + 3) / 4) : Ho * Wo, C).nslab;
}

int mve_seg_dwconv_pool(int dtype, const void* x, int B, int H, int W, int C, int ldx, const float* w, const float* bias, void* out, int Ho, int Wo,
                        int ldo, int kh, int kw, int stride, int pad_t, int pad_l, int act, float* sums, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(x && w && out && sums && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && kh > 0 && kw > 0 && stride > 0, MVE_ERR_ARG,
              "seg_dwconv_pool: channel count / strides must be multiples of 8");
    ConvD p;
    p.x = x; p.w = w; p.bias = bias; p.out = out; p.add = nullptr; p.mul = nullptr;
    p.B = B; p.H = H; p.W = W; p.Cin = C; p.ldx = ldx; p.Ho = Ho; p.Wo = Wo; p.Cout = C; p.ldo = ldo; p.kh = kh; p.kw = kw;
    p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l; p.dil = 1; p.act = act; p.ld2 = 0;
    if (dtype == MVE_F16) return conv2d_run<F16Tag>(p, 1, 0, (hipStream_t)stream, sums);
    if (dtype == MVE_BF16) return conv2d_run<BF16Tag>(p, 1, 0, (hipStream_t)stream, sums);
    mve_set_error("seg_dwconv_pool: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

int mve_seg_scale(int dtype, void* x, const void* src, int B, int HW, int C, const float* A, const float* S, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(x && src && A && C % 8 == 0, MVE_ERR_ARG, "seg_scale: C must be a multiple of 8");
    const size_t n8 = (size_t)B * HW * (C / 8);
    const unsigned grid = mve_cdiv(n8, 256);
    if (dtype == MVE_F16) k_seg_scale<F16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(x, src, HW, C, A, S, n8);
    else if (dtype == MVE_BF16) k_seg_scale<BF16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(x, src, HW, C, A, S, n8);
    else { mve_set_error("seg_scale: unsupported dtype %d", dtype); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_seg_resize(int dtype, const void* x, int B, int H, int W, int C, void* out, int Ho, int Wo, int align_corners, int in_mode, int out_f32,
                   const float* mean, const float* stdv, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(x && out && H > 0 && W > 0 && Ho > 0 && Wo > 0 && in_mode >= 0 && in_mode <= 2, MVE_ERR_ARG, "seg_resize: bad arguments");
    MVE_CHECK((mean == nullptr) == (stdv == nullptr), MVE_ERR_ARG, "seg_resize: mean and std go together");
    if (dtype == MVE_F16) return resize_run<F16Tag>(x, B, H, W, C, out, Ho, Wo, align_corners, in_mode, out_f32, mean, stdv, (hipStream_t)stream);
    if (dtype == MVE_BF16) return resize_run<BF16Tag>(x, B, H, W, C, out, Ho, Wo, align_corners, in_mode, out_f32, mean, stdv, (hipStream_t)stream);
    mve_set_error("seg_resize: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

int mve_seg_uam_channel(const float* pooled, int B, int C, const float* ns, const float* nb, const float* wq, const float* wk, const float* wv,
                        const float* wfc, float ratio, const float* bs, const float* bt, float* att, float* A, float* S, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(pooled && ns && nb && wq && wk && wv && wfc && bs && bt && att && A && S && C > 0 && C <= 256, MVE_ERR_ARG, "seg_uam_channel: C must be <= 256");
    k_seg_uam_channel<<<B, 256, 0, (hipStream_t)stream>>>(pooled, C, ns, nb, wq, wk, wv, wfc, ratio, bs, bt, att, A, S);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_seg_uam_spatial(const float* qkv, int B, int H, int W, float* out, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(qkv && out && H > 0 && W > 0 && (size_t)(H + W) * 4 <= 60000, MVE_ERR_ARG, "seg_uam_spatial: map too large");
    dim3 grid(H, B);
    k_seg_uam_spatial<<<grid, 256, (H + W) * sizeof(float), (hipStream_t)stream>>>(qkv, H, W, out);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_seg_object_mix(int dtype, const float* d, const void* enc, void* out, int B, int HW, int C, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d && enc && out && C % 8 == 0, MVE_ERR_ARG, "seg_object_mix: C must be a multiple of 8");
    const size_t n8 = (size_t)B * HW * (C / 8);
    const unsigned grid = mve_cdiv(n8, 256);
    if (dtype == MVE_F16) k_seg_object_mix<F16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(d, enc, out, C, n8);
    else if (dtype == MVE_BF16) k_seg_object_mix<BF16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(d, enc, out, C, n8);
    else { mve_set_error("seg_object_mix: unsupported dtype %d", dtype); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_seg_mul(int dtype, const void* x, const void* y, const void* z, void* out, size_t n, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(x && y && out && n % 8 == 0, MVE_ERR_ARG, "seg_mul: n must be a multiple of 8");
    const unsigned grid = mve_cdiv(n / 8, 256);
    if (dtype == MVE_F16) k_seg_mul<F16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(x, y, z, out, n / 8);
    else if (dtype == MVE_BF16) k_seg_mul<BF16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(x, y, z, out, n / 8);
    else { mve_set_error("seg_mul: unsupported dtype %d", dtype); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_seg_fuse(const float* d0, const float* d1, const float* d2, int B, int Hs, int Ws, float* out, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d0 && d1 && d2 && out && Hs % 8 == 0 && Ws % 8 == 0, MVE_ERR_ARG, "seg_fuse: the input size must be a multiple of 8");
    k_seg_fuse<<<mve_cdiv((size_t)B * Hs * Ws, 256), 256, 0, (hipStream_t)stream>>>(d0, d1, d2, B, Hs, Ws, out);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

// workspace: B * Hs * Ws + B * Ho * Wo floats + B ints
size_t mve_seg_post_workspace_bytes(int B, int Hs, int Ws, int Ho, int Wo) {
    return ((size_t)B * Hs * Ws + (size_t)B * Ho * Wo) * sizeof(float) + (size_t)(B + 16) * sizeof(int);
}

int mve_seg_post(int dtype, const float* m, int B, int Hs, int Ws, int erosion, void* out, int Ho, int Wo, int out_f32, void* workspace, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(m && out && workspace && erosion >= 0, MVE_ERR_ARG, "seg_post: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float* er = reinterpret_cast<float*>(workspace);
    float* rs = er + (size_t)B * Hs * Ws;
    int* flag = reinterpret_cast<int*>(rs + (size_t)B * Ho * Wo);
    MVE_HIP(hipMemsetAsync(flag, 0, B * sizeof(int), s));
    k_seg_erode<<<mve_cdiv((size_t)B * Hs * Ws, 256), 256, 0, s>>>(m, B, Hs, Ws, erosion, er);
    MVE_LAUNCH_CHECK();
    const unsigned grid = mve_cdiv((size_t)B * Ho * Wo, 256);
    if (dtype == MVE_F16) {
        k_seg_post_resize<F16Tag><<<grid, 256, 0, s>>>(er, B, Hs, Ws, Ho, Wo, rs, flag, out_f32);
        k_seg_post_apply<F16Tag><<<grid, 256, 0, s>>>(rs, B, Ho * Wo, flag, out, out_f32);
    } else if (dtype == MVE_BF16) {
        k_seg_post_resize<BF16Tag><<<grid, 256, 0, s>>>(er, B, Hs, Ws, Ho, Wo, rs, flag, out_f32);
        k_seg_post_apply<BF16Tag><<<grid, 256, 0, s>>>(rs, B, Ho * Wo, flag, out, out_f32);
    } else { mve_set_error("seg_post: unsupported dtype %d", dtype); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
