// Kernels of the TRACER-B7 foreground segmentor (lib/models/segmentors/tracer_b7.py:16-73 over lib/models/architecture/tracerb7/: EfficientNet-B7
// encoder + TRACER decoder), the pieces that the GEMM / 3x3-conv kernels do not cover.  Activations are NHWC 16-bit ([B, H, W, C] contiguous, the
// layout every 1x1 convolution consumes as a plain [M, C] GEMM operand); single-channel decoder maps are fp32.  BatchNorm is folded into the
// convolution weights and a per-channel bias by the host mirror (mvedit_amd/segmentor.py) when the state dict is loaded.  All of these are
// HBM- / L2-bound gathers with fp32 arithmetic; nothing here is GEMM-shaped enough to go to the matrix cores except what already does
// (expand / project 1x1 convolutions -> mve_gemm, dense 3x3 convolutions of the decoder -> mve_conv3x3).
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

// depthwise: one thread per (output pixel, 8 consecutive channels); weights [kh][kw][C] fp32
template <class Tag>
__global__ __launch_bounds__(256) void k_seg_dwconv(const ConvD p) {
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    const int c8n = p.Cin / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.Ho * p.Wo * c8n;
    if (idx >= total) return;
    const int c0 = (int)(idx % c8n) * 8;
    const long long pix = idx / c8n;
    const int xo = (int)(pix % p.Wo), yo = (int)((pix / p.Wo) % p.Ho), b = (int)(pix / ((long long)p.Wo * p.Ho));
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = p.bias ? p.bias[c0 + e] : 0.f;
    const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)b * p.H * p.W * p.ldx + c0;
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
    for (int e = 0; e < 8; ++e) o[e] = Tag::from_f32(seg_act(acc[e], p.act));
    *reinterpret_cast<V8*>(reinterpret_cast<T*>(p.out) + (size_t)pix * p.ldo + c0) = o;
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

// gate[b][c] = sigmoid(W2[c][:] . swish(W1 pooled[b] + b1) + b2[c]); one block per image
__global__ __launch_bounds__(256) void k_seg_se_gate(const float* pooled, int C, int S, const float* w1, const float* b1, const float* w2, const float* b2,
                                                     float* gate) {
    extern __shared__ float sh[];          // S hidden values
    const int b = blockIdx.x;
    const float* pv = pooled + (size_t)b * C;
    for (int s = threadIdx.x; s < S; s += 256) {
        float a = b1[s];
        for (int c = 0; c < C; ++c) a = __builtin_fmaf(w1[(size_t)s * C + c], pv[c], a);
        sh[s] = a / (1.0f + __expf(-a));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = b2[c];
        for (int s = 0; s < S; ++s) a = __builtin_fmaf(w2[(size_t)c * S + s], sh[s], a);
        gate[(size_t)b * C + c] = 1.0f / (1.0f + __expf(-a));
    }
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
int conv2d_run(const ConvD& p, int depthwise, int out_f32, hipStream_t s) {
    if (depthwise) {
        const long long total = (long long)p.B * p.Ho * p.Wo * (p.Cin / 8);
        k_seg_dwconv<Tag><<<mve_cdiv(total, 256), 256, 0, s>>>(p);
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

int mve_seg_se_gate(const float* pooled, int B, int C, int S, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                    void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(pooled && w1 && b1 && w2 && b2 && gate && S > 0 && S <= 4096, MVE_ERR_ARG, "seg_se_gate: bad arguments");
    k_seg_se_gate<<<B, 256, S * sizeof(float), (hipStream_t)stream>>>(pooled, C, S, w1, b1, w2, b2, gate);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
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
