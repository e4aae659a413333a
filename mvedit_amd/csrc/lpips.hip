// LPIPS-VGG building blocks that are not GEMMs (gfx950; all HBM-bound, NHWC 16-bit activations like the UNet's):
// input scaling, 2x2 max pooling and its backward, ReLU backward, and the per-layer perceptual distance with its backward.
// They surround the VGG16 convolutions (csrc/gemm.hip) in the executor's LPIPS mode (csrc/unet.hip, Config::lpips), which replaces
// `lpips.LPIPS(net='vgg')` as the reference calls it for its patch loss (lib/models/losses/lpips_loss.py:8-42, used by
// lib/models/autoencoders/base_nerf.py:337-344 on 128 x 128 patches every optimisation iteration), forward AND backward w.r.t. the
// prediction.  lpips==0.1.4 (requirements.txt) semantics:
//   in  = ((2 x - 1) - shift) / scale                        ScalingLayer (after LPIPSLoss.normalize_inputs)
//   f_l = VGG16 relu{1_2, 2_2, 3_3, 4_3, 5_3}(in)
//   u   = f / (sqrt(sum_c f^2) + 1e-10)                       normalize_tensor
//   d_l = mean_{h,w} sum_c w_{l,c} (u_pred - u_target)^2      lin layers (1x1 conv, no bias) + spatial_average
//   loss[n] = sum_l d_l[n]
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr float LPIPS_EPS = 1e-10f;

template <class Tag, class Src>
__global__ __launch_bounds__(NT) void k_lpips_scale(const Src* __restrict__ pred, const Src* __restrict__ target, int B, int HW,
                                                    const float* __restrict__ shift, const float* __restrict__ scale, int normalize,
                                                    typename Tag::T* __restrict__ out /* [2B*HW][8] */) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;          // over 2B*HW pixels
    if (i >= (size_t)2 * B * HW) return;
    const int img = (int)(i / HW), r = (int)(i - (size_t)img * HW);
    const Src* src = img < B ? pred + (size_t)img * 3 * HW : target + (size_t)(img - B) * 3 * HW;
    V8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = Tag::from_f32(0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = (float)src[(size_t)c * HW + r];
        if (normalize) v = v * 2.0f - 1.0f;
        o[c] = Tag::from_f32((v - shift[c]) / scale[c]);
    }
    reinterpret_cast<V8*>(out)[i] = o;
}

// d pred[b][c][y][x] = g[(b,y,x)][c] * (normalize ? 2 : 1) / scale_c
template <class Tag, class Dst>
__global__ __launch_bounds__(NT) void k_lpips_input_grad(const typename Tag::T* __restrict__ g, int B, int HW, const float* __restrict__ scale,
                                                         int normalize, Dst* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;          // over B*3*HW (output order)
    if (i >= (size_t)B * 3 * HW) return;
    const int r = (int)(i % HW);
    const size_t t = i / HW;
    const int c = (int)(t % 3), b = (int)(t / 3);
    out[i] = (Dst)(Tag::to_f32(g[((size_t)b * HW + r) * 8 + c]) / scale[c] * (normalize ? 2.0f : 1.0f));
}

// MaxPool2d(2, 2) on NHWC, 8 channels per thread
template <class Tag>
__global__ __launch_bounds__(NT) void k_maxpool(const typename Tag::T* __restrict__ x, int B, int H, int W, int C8, typename Tag::T* __restrict__ y) {
    typedef typename Tag::V8 V8;
    const int Ho = H / 2, Wo = W / 2;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;          // over B*Ho*Wo*C8
    if (i >= (size_t)B * Ho * Wo * C8) return;
    const int c = (int)(i % C8);
    size_t t = i / C8;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const V8* xin = reinterpret_cast<const V8*>(x);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const V8 v = xin[(((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C8 + c];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], Tag::to_f32(v[e]));
    }
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = Tag::from_f32(m[e]);
    reinterpret_cast<V8*>(y)[i] = o;
}

// backward of MaxPool2d(2, 2): the gradient of an output goes to the FIRST window element (row-major) that equals the maximum
// (torch's argmax rule); every input element is written (zero where it was not the arg-max)
template <class Tag>
__global__ __launch_bounds__(NT) void k_maxpool_bwd(const typename Tag::T* __restrict__ x, const typename Tag::T* __restrict__ gy, int B, int H,
                                                    int W, int C8, typename Tag::T* __restrict__ gx) {
    typedef typename Tag::V8 V8;
    const int Ho = H / 2, Wo = W / 2;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;          // over B*Ho*Wo*C8 (one thread per window)
    if (i >= (size_t)B * Ho * Wo * C8) return;
    const int c = (int)(i % C8);
    size_t t = i / C8;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const V8* xin = reinterpret_cast<const V8*>(x);
    V8 v[4];
    float m[8];
    int arg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; arg[e] = 0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = xin[(((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C8 + c];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = Tag::to_f32(v[k][e]);
            if (f > m[e]) { m[e] = f; arg[e] = k; }
        }
    }
    const V8 g = reinterpret_cast<const V8*>(gy)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = arg[e] == k ? g[e] : Tag::from_f32(0.f);
        reinterpret_cast<V8*>(gx)[(((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C8 + c] = o;
    }
}

// g = a > 0 ? g : 0   (threshold_backward of ReLU, on the saved OUTPUT as torch does)
template <class Tag>
__global__ __launch_bounds__(NT) void k_relu_bwd(typename Tag::T* __restrict__ g, const typename Tag::T* __restrict__ a, size_t n8) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n8) return;
    V8 gv = reinterpret_cast<V8*>(g)[i];
    const V8 av = reinterpret_cast<const V8*>(a)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) if (!(Tag::to_f32(av[e]) > 0.f)) gv[e] = Tag::from_f32(0.f);
    reinterpret_cast<V8*>(g)[i] = gv;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// One wave per pixel, lanes stride the channel chunks.  Forward: partial[n][blk] = sum over the block's pixels of
// sum_c w_c (u0_c - u1_c)^2;  MODE 1 (backward): g[(n,p)][c] = coef_n * (2 w_c d_c / n0 - (sum_k 2 w_k d_k f0_k) f0_c / (n0^2 |f0|)).
template <class Tag, int MODE>
__global__ __launch_bounds__(NT) void k_lpips_layer(const typename Tag::T* __restrict__ f, const float* __restrict__ w, int B, int HW, int C,
                                                    int pix_per_block, float* __restrict__ partial, const float* __restrict__ grad_out,
                                                    typename Tag::T* __restrict__ g) {
    // no fma contraction in this kernel: d = x i0 - y i1 must round both products before the subtraction, so that identical features give
    // d = 0 exactly as in the reference (first GPU run: a contracted fma left 2e-17 for identical images); the kernel is HBM-bound
#pragma clang fp contract(off)
    typedef typename Tag::V8 V8;
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y, C8 = C / 8;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    const V8* f0 = reinterpret_cast<const V8*>(f) + (size_t)n * HW * C8;
    const V8* f1 = reinterpret_cast<const V8*>(f) + (size_t)(B + n) * HW * C8;
    float acc = 0.f;
    float coef = 0.f;
    if constexpr (MODE == 1) coef = grad_out[n] / (float)HW;
    for (int p = p0 + wave; p < p1; p += 4) {
        float s0 = 0.f, s1 = 0.f;
        for (int c = lane; c < C8; c += 64) {
            const V8 a = f0[(size_t)p * C8 + c], b = f1[(size_t)p * C8 + c];
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float x = Tag::to_f32(a[e]), y = Tag::to_f32(b[e]); s0 += x * x; s1 += y * y; }
        }
        s0 = wave_sum(s0); s1 = wave_sum(s1);
        const float r0 = sqrtf(s0), r1 = sqrtf(s1);
        const float i0 = 1.0f / (r0 + LPIPS_EPS), i1 = 1.0f / (r1 + LPIPS_EPS);
        float val = 0.f, dot = 0.f;
        for (int c = lane; c < C8; c += 64) {
            const V8 a = f0[(size_t)p * C8 + c], b = f1[(size_t)p * C8 + c];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = Tag::to_f32(a[e]);
                const float d = x * i0 - Tag::to_f32(b[e]) * i1, wc = w[c * 8 + e];
                val += wc * d * d;
                dot += 2.0f * wc * d * x;
            }
        }
        if constexpr (MODE == 0) {
            acc += wave_sum(val);
        } else {
            dot = wave_sum(dot);
            const float k2 = r0 > 0.f ? dot * i0 * i0 / r0 : 0.f;          // d(1 / (|f| + eps)) / d f_c = -f_c / (|f| (|f| + eps)^2)
            V8* go = reinterpret_cast<V8*>(g) + ((size_t)n * HW + p) * C8;
            for (int c = lane; c < C8; c += 64) {
                const V8 a = f0[(size_t)p * C8 + c], b = f1[(size_t)p * C8 + c];
                V8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = Tag::to_f32(a[e]);
                    const float d = x * i0 - Tag::to_f32(b[e]) * i1;
                    o[e] = Tag::from_f32(coef * (2.0f * w[c * 8 + e] * d * i0 - k2 * x));
                }
                go[c] = o;
            }
        }
    }
    if constexpr (MODE == 0) {
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// loss[n] (+)= (sum_blk partial[n][blk]) / HW, fixed order
__global__ void k_lpips_reduce(const float* __restrict__ partial, int nblk, int HW, int accumulate, float* __restrict__ loss) {
    const int n = blockIdx.x;
    if (threadIdx.x != 0) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += partial[(size_t)n * nblk + i];
    s /= (float)HW;
    loss[n] = accumulate ? loss[n] + s : s;
}

int lpips_blocks(int HW) { const int ppb = 64; return (HW + ppb - 1) / ppb; }

}  // namespace

extern "C" {

int mve_lpips_scale(int dtype, int io_dtype, const void* d_pred, const void* d_target, int B, int H, int W, const float* d_shift3,
                    const float* d_scale3, int normalize, void* d_out, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_pred && d_target && d_out && d_shift3 && d_scale3, MVE_ERR_ARG, "lpips_scale: null pointer");
    const float* sh = d_shift3; const float* is = d_scale3;
    const int HW = H * W;
    const unsigned grid = mve_cdiv((size_t)2 * B * HW, NT);
    hipStream_t s = (hipStream_t)stream;
#define MVE_LPIPS_SCALE(TAG)                                                                                                           \
    if (io_dtype == MVE_F32) k_lpips_scale<TAG, float><<<grid, NT, 0, s>>>((const float*)d_pred, (const float*)d_target, B, HW, sh, is, normalize, (typename TAG::T*)d_out); \
    else if (io_dtype == MVE_F16) k_lpips_scale<TAG, f16><<<grid, NT, 0, s>>>((const f16*)d_pred, (const f16*)d_target, B, HW, sh, is, normalize, (typename TAG::T*)d_out);  \
    else if (io_dtype == MVE_BF16) k_lpips_scale<TAG, bf16><<<grid, NT, 0, s>>>((const bf16*)d_pred, (const bf16*)d_target, B, HW, sh, is, normalize, (typename TAG::T*)d_out); \
    else { mve_set_error("lpips_scale: bad io dtype"); return MVE_ERR_ARG; }
    if (dtype == MVE_F16) { MVE_LPIPS_SCALE(F16Tag) }
    else if (dtype == MVE_BF16) { MVE_LPIPS_SCALE(BF16Tag) }
    else { mve_set_error("lpips_scale: bad dtype"); return MVE_ERR_ARG; }
#undef MVE_LPIPS_SCALE
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_lpips_input_grad(int dtype, int io_dtype, const void* d_g8, int B, int H, int W, const float* d_scale3, int normalize, void* d_out,
                         void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_g8 && d_out && d_scale3, MVE_ERR_ARG, "lpips_input_grad: null pointer");
    const float* is = d_scale3;
    const int HW = H * W;
    const unsigned grid = mve_cdiv((size_t)B * 3 * HW, NT);
    hipStream_t s = (hipStream_t)stream;
#define MVE_LPIPS_IG(TAG)                                                                                                        \
    if (io_dtype == MVE_F32) k_lpips_input_grad<TAG, float><<<grid, NT, 0, s>>>((const typename TAG::T*)d_g8, B, HW, is, normalize, (float*)d_out); \
    else if (io_dtype == MVE_F16) k_lpips_input_grad<TAG, f16><<<grid, NT, 0, s>>>((const typename TAG::T*)d_g8, B, HW, is, normalize, (f16*)d_out);  \
    else if (io_dtype == MVE_BF16) k_lpips_input_grad<TAG, bf16><<<grid, NT, 0, s>>>((const typename TAG::T*)d_g8, B, HW, is, normalize, (bf16*)d_out); \
    else { mve_set_error("lpips_input_grad: bad io dtype"); return MVE_ERR_ARG; }
    if (dtype == MVE_F16) { MVE_LPIPS_IG(F16Tag) }
    else if (dtype == MVE_BF16) { MVE_LPIPS_IG(BF16Tag) }
    else { mve_set_error("lpips_input_grad: bad dtype"); return MVE_ERR_ARG; }
#undef MVE_LPIPS_IG
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_maxpool2x2(int dtype, const void* d_x, int B, int H, int W, int C, void* d_y, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_x && d_y && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, MVE_ERR_ARG, "maxpool2x2: even H, W and C %% 8 == 0 required");
    const unsigned grid = mve_cdiv((size_t)B * (H / 2) * (W / 2) * (C / 8), NT);
    if (dtype == MVE_F16) k_maxpool<F16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const f16*)d_x, B, H, W, C / 8, (f16*)d_y);
    else if (dtype == MVE_BF16) k_maxpool<BF16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const bf16*)d_x, B, H, W, C / 8, (bf16*)d_y);
    else { mve_set_error("maxpool2x2: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_maxpool2x2_backward(int dtype, const void* d_x, const void* d_grad_y, int B, int H, int W, int C, void* d_grad_x, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_x && d_grad_y && d_grad_x && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, MVE_ERR_ARG, "maxpool2x2_backward: even H, W and C %% 8 == 0 required");
    const unsigned grid = mve_cdiv((size_t)B * (H / 2) * (W / 2) * (C / 8), NT);
    if (dtype == MVE_F16) k_maxpool_bwd<F16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const f16*)d_x, (const f16*)d_grad_y, B, H, W, C / 8, (f16*)d_grad_x);
    else if (dtype == MVE_BF16) k_maxpool_bwd<BF16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const bf16*)d_x, (const bf16*)d_grad_y, B, H, W, C / 8, (bf16*)d_grad_x);
    else { mve_set_error("maxpool2x2_backward: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_relu_backward(int dtype, void* d_grad, const void* d_out, size_t n, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(d_grad && d_out && n % 8 == 0, MVE_ERR_ARG, "relu_backward: n must be a multiple of 8");
    const unsigned grid = mve_cdiv(n / 8, NT);
    if (dtype == MVE_F16) k_relu_bwd<F16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((f16*)d_grad, (const f16*)d_out, n / 8);
    else if (dtype == MVE_BF16) k_relu_bwd<BF16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((bf16*)d_grad, (const bf16*)d_out, n / 8);
    else { mve_set_error("relu_backward: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

size_t mve_lpips_layer_scratch_bytes(int B, int HW) { return (size_t)(B > 0 ? B : 0) * lpips_blocks(HW) * sizeof(float) + 64; }

int mve_lpips_layer(int dtype, const void* d_feat, const float* d_lin_w, int B, int HW, int C, int accumulate, float* d_loss, void* d_scratch,
                    void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_feat && d_lin_w && d_loss && d_scratch && C % 8 == 0 && HW > 0, MVE_ERR_ARG, "lpips_layer: bad arguments");
    const int nblk = lpips_blocks(HW);
    dim3 grid(nblk, B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MVE_F16) k_lpips_layer<F16Tag, 0><<<grid, NT, 0, s>>>((const f16*)d_feat, d_lin_w, B, HW, C, 64, (float*)d_scratch, nullptr, nullptr);
    else if (dtype == MVE_BF16) k_lpips_layer<BF16Tag, 0><<<grid, NT, 0, s>>>((const bf16*)d_feat, d_lin_w, B, HW, C, 64, (float*)d_scratch, nullptr, nullptr);
    else { mve_set_error("lpips_layer: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    k_lpips_reduce<<<B, 64, 0, s>>>((const float*)d_scratch, nblk, HW, accumulate, d_loss);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_lpips_layer_backward(int dtype, const void* d_feat, const float* d_lin_w, const float* d_grad_loss, int B, int HW, int C,
                             void* d_grad_feat, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_feat && d_lin_w && d_grad_loss && d_grad_feat && C % 8 == 0 && HW > 0, MVE_ERR_ARG, "lpips_layer_backward: bad arguments");
    dim3 grid(lpips_blocks(HW), B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MVE_F16) k_lpips_layer<F16Tag, 1><<<grid, NT, 0, s>>>((const f16*)d_feat, d_lin_w, B, HW, C, 64, nullptr, d_grad_loss, (f16*)d_grad_feat);
    else if (dtype == MVE_BF16) k_lpips_layer<BF16Tag, 1><<<grid, NT, 0, s>>>((const bf16*)d_feat, d_lin_w, B, HW, C, 64, nullptr, d_grad_loss, (bf16*)d_grad_feat);
    else { mve_set_error("lpips_layer_backward: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
