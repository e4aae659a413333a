// Small HBM-bound helpers around the UNet hot path (gfx950): layout conversion at the NCHW boundary of the
// reference's `unet(sample, ...)` seam, the sinusoidal timestep embedding, SiLU, residual adds and the
// classifier-free-guidance combine of lib/pipelines/adapter3d_mixin.py:129-134.
#include "common.h"

namespace {

constexpr int NT = 256;

// NCHW (f32 | f16 | bf16) -> NHWC 16-bit with the channel axis zero-padded to Cpad (multiple of 8)
template <class Tag, class Src>
__global__ __launch_bounds__(NT) void k_nchw_to_nhwc(const Src* __restrict__ x, int B, int C, int HW, int Cpad,
                                                     typename Tag::T* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;   // over B*HW*Cpad
    const size_t total = (size_t)B * HW * Cpad;
    if (i >= total) return;
    const int c = (int)(i % Cpad);
    const size_t pix = i / Cpad;
    const int b = (int)(pix / HW), r = (int)(pix - (size_t)b * HW);
    float v = 0.f;
    if (c < C) v = (float)x[((size_t)b * C + c) * HW + r];
    y[i] = Tag::from_f32(v);
}

// NHWC (16-bit or f32, row stride ld) -> NCHW (f32 | 16-bit), first C channels
template <class SrcT, class Dst>
__global__ __launch_bounds__(NT) void k_nhwc_to_nchw(const SrcT* __restrict__ x, int ld, int B, int C, int HW,
                                                     Dst* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;   // over B*C*HW (output order)
    const size_t total = (size_t)B * C * HW;
    if (i >= total) return;
    const int r = (int)(i % HW);
    const size_t bc = i / HW;
    const int c = (int)(bc % C), b = (int)(bc / C);
    y[i] = (Dst)(float)x[((size_t)b * HW + r) * ld + c];
}

// diffusers Timesteps(num_channels=dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] halves
template <class Tag>
__global__ void k_timestep_embedding(const float* __restrict__ t, int B, int dim, typename Tag::T* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, c = i - b * dim, half = dim / 2;
    const int k = c < half ? c : c - half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);   // ln(10000)
    const float arg = t[b] * freq;
    out[i] = Tag::from_f32(c < half ? cosf(arg) : sinf(arg));
}

template <class Tag>
__global__ __launch_bounds__(NT) void k_silu(const typename Tag::T* __restrict__ x, typename Tag::T* __restrict__ y, size_t n8) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n8) return;
    const V8 v = reinterpret_cast<const V8*>(x)[i];
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = Tag::to_f32(v[e]); o[e] = Tag::from_f32(f / (1.0f + __expf(-f))); }
    reinterpret_cast<V8*>(y)[i] = o;
}

// y = a + alpha * b   (16-bit, fp32 math)
template <class Tag>
__global__ __launch_bounds__(NT) void k_axpy(const typename Tag::T* __restrict__ a, const typename Tag::T* __restrict__ b,
                                             float alpha, typename Tag::T* __restrict__ y, size_t n8) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n8) return;
    const V8 va = reinterpret_cast<const V8*>(a)[i], vb = reinterpret_cast<const V8*>(b)[i];
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = Tag::from_f32(Tag::to_f32(va[e]) + alpha * Tag::to_f32(vb[e]));
    reinterpret_cast<V8*>(y)[i] = o;
}

// (y, y_lo) = pair of (a + a_lo) + alpha * b: a stream tensor of the executor's residual_pair mode plus a 16-bit addend; the low halves are lo8
// (one byte per element, common.h)
template <class Tag>
__global__ __launch_bounds__(NT) void k_axpy_pair(const typename Tag::T* __restrict__ a, const unsigned char* __restrict__ al,
                                                  const typename Tag::T* __restrict__ b, float alpha, typename Tag::T* __restrict__ y,
                                                  unsigned char* __restrict__ yl, size_t n8) {
#pragma clang fp contract(off)
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n8) return;
    const V8 va = reinterpret_cast<const V8*>(a)[i], vb = reinterpret_cast<const V8*>(b)[i];
    u32x2 vl = {0u, 0u};
    if (al) vl = reinterpret_cast<const u32x2*>(al)[i];
    float s[8], v[8];
    mve_pair_load8<Tag>(va, vl, s);
    bool nothing = true;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float add = alpha * Tag::to_f32(vb[e]);
        nothing = nothing && add == 0.f;
        v[e] = s[e] + add;
    }
    V8 o;
    u32x2 ol = mve_pair_split8<Tag>(v, o);
    // adding nothing must change nothing, bit for bit (unet_dec without ControlNet feeds zero residuals through this kernel): re-splitting
    // hi + lo can move hi by one step when |lo| was rounded up to exactly half a step of an odd hi
    if (nothing) { o = va; ol = vl; }
    reinterpret_cast<V8*>(y)[i] = o;
    reinterpret_cast<u32x2*>(yl)[i] = ol;
}

// noise = g * text + (1 - g) * uncond   on f32 NCHW latents (adapter3d_mixin.py:130-134)
__global__ __launch_bounds__(NT) void k_cfg(const float* __restrict__ uncond, const float* __restrict__ text, float g,
                                            float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) out[i] = g * text[i] + (1.0f - g) * uncond[i];
}

}  // namespace

namespace {
// pred_original_sample of the reference's denoise loop (lib/pipelines/mvedit_3d_pipeline.py:1253-1255)
__global__ __launch_bounds__(256) void k_x0_prediction(const float* __restrict__ x, const float* __restrict__ e, float sa, float sb, size_t n,
                                                       float* __restrict__ out) {
#pragma clang fp contract(off)      // no fma: bit-equal to the host expression evaluated op by op
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (x[i] - sb * e[i]) / sa;
}
}  // namespace

namespace {
// Row softmax of fp32 scores into the 16-bit probabilities the P.V GEMM consumes (single-head attention of the VAE mid block,
// whose head dim of 512 is outside the fused attention kernel's range): one block per row, fp32 max / sum, exp2 arithmetic.
template <class Tag>
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ S, size_t lds, int N, typename Tag::T* __restrict__ P,
                                                      size_t ldp) {
    __shared__ float red[4];
    const float* s = S + (size_t)blockIdx.x * lds;
    typename Tag::T* o = P + (size_t)blockIdx.x * ldp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto all_max = [&](float v) {
        for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    };
    auto all_sum = [&](float v) {
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    float mx = -INFINITY;
    for (int i = threadIdx.x * 4; i < N; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + i);
        mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    mx = all_max(mx);
    constexpr float LOG2E = 1.44269504088896340736f;
    float sum = 0.f;
    for (int i = threadIdx.x * 4; i < N; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + i);
        sum += (exp2f((v[0] - mx) * LOG2E) + exp2f((v[1] - mx) * LOG2E)) + (exp2f((v[2] - mx) * LOG2E) + exp2f((v[3] - mx) * LOG2E));
    }
    sum = all_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = threadIdx.x * 4; i < N; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + i);
        typedef typename Tag::T T4 __attribute__((ext_vector_type(4)));
        T4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = Tag::from_f32(exp2f((v[e] - mx) * LOG2E) * inv);
        *reinterpret_cast<T4*>(o + i) = r;
    }
}
}  // namespace

namespace {
// per-channel PReLU on NHWC rows (nn.PReLU(num_parameters=C) of SRVGGNetCompact, lib/models/decoders/image_space_ss.py:41-56)
template <class Tag>
__global__ __launch_bounds__(NT) void k_prelu(const typename Tag::T* __restrict__ x, const float* __restrict__ slope, int c8,
                                              typename Tag::T* __restrict__ y, size_t n8) {
    typedef typename Tag::V8 V8;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n8) return;
    const int c0 = (int)(i % (size_t)c8) * 8;
    const V8 v = reinterpret_cast<const V8*>(x)[i];
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(slope + c0), a1 = *reinterpret_cast<const f32x4*>(slope + c0 + 4);
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float f = Tag::to_f32(v[e]);
        o[e] = Tag::from_f32(f >= 0.f ? f : f * (e < 4 ? a0[e] : a1[e - 4]));
    }
    reinterpret_cast<V8*>(y)[i] = o;
}

// nn.PixelShuffle(r) of an NHWC fp32 tensor + the nearest-upsampled network input (image_space_ss.py:63-70):
//   out[b][c][y*r+i][x*r+j] = src[(b,y,x)][c*r*r + i*r + j] + base[b][c][y][x],  NCHW output in the caller's dtype
template <class Dst, class Base>
__global__ __launch_bounds__(NT) void k_pixel_shuffle_add(const float* __restrict__ src, int ld, const Base* __restrict__ base, int B, int C,
                                                          int H, int W, int r, Dst* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;          // over the output, x fastest
    const int Wo = W * r, Ho = H * r;
    const size_t total = (size_t)B * C * Ho * Wo;
    if (i >= total) return;
    const int X = (int)(i % Wo);
    const size_t t = i / Wo;
    const int Y = (int)(t % Ho);
    const size_t u = t / Ho;
    const int c = (int)(u % C), b = (int)(u / C);
    const int y = Y / r, x = X / r, ii = Y - y * r, jj = X - x * r;
    const float v = src[(((size_t)b * H + y) * W + x) * ld + c * r * r + ii * r + jj] + (float)base[(((size_t)b * C + c) * H + y) * W + x];
    out[i] = (Dst)v;
}

template <class Dst>
int pixel_shuffle_add_dst(int base_dtype, const float* src, int ld, const void* base, int B, int C, int H, int W, int r, Dst* out, hipStream_t s) {
    const unsigned grid = mve_cdiv((size_t)B * C * H * r * W * r, NT);
    if (base_dtype == MVE_F32) k_pixel_shuffle_add<Dst, float><<<grid, NT, 0, s>>>(src, ld, (const float*)base, B, C, H, W, r, out);
    else if (base_dtype == MVE_F16) k_pixel_shuffle_add<Dst, f16><<<grid, NT, 0, s>>>(src, ld, (const f16*)base, B, C, H, W, r, out);
    else if (base_dtype == MVE_BF16) k_pixel_shuffle_add<Dst, bf16><<<grid, NT, 0, s>>>(src, ld, (const bf16*)base, B, C, H, W, r, out);
    else { mve_set_error("pixel_shuffle_add: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}
}  // namespace

extern "C" {

int mve_nchw_to_nhwc(int dst_dtype, int src_dtype, const void* x, int B, int C, int H, int W, int Cpad, void* y, void* stream) {
    MVE_CHECK(Cpad >= C && Cpad % 8 == 0, MVE_ERR_ARG, "nchw_to_nhwc: Cpad=%d must be >= C=%d and a multiple of 8", Cpad, C);
    const size_t total = (size_t)B * H * W * Cpad;
    if (total == 0) return MVE_OK;
    MVE_CHECK(x && y, MVE_ERR_ARG, "nchw_to_nhwc: null pointer");
    const unsigned grid = mve_cdiv(total, NT);
    hipStream_t s = (hipStream_t)stream;
#define GO(TAG, SRC) k_nchw_to_nhwc<TAG, SRC><<<grid, NT, 0, s>>>((const SRC*)x, B, C, H * W, Cpad, (typename TAG::T*)y)
    if (dst_dtype == MVE_F16) {
        if (src_dtype == MVE_F32) GO(F16Tag, float);
        else if (src_dtype == MVE_F16) GO(F16Tag, f16);
        else if (src_dtype == MVE_BF16) GO(F16Tag, bf16);
        else { mve_set_error("nchw_to_nhwc: bad src dtype %d", src_dtype); return MVE_ERR_ARG; }
    } else if (dst_dtype == MVE_BF16) {
        if (src_dtype == MVE_F32) GO(BF16Tag, float);
        else if (src_dtype == MVE_F16) GO(BF16Tag, f16);
        else if (src_dtype == MVE_BF16) GO(BF16Tag, bf16);
        else { mve_set_error("nchw_to_nhwc: bad src dtype %d", src_dtype); return MVE_ERR_ARG; }
    } else { mve_set_error("nchw_to_nhwc: bad dst dtype %d", dst_dtype); return MVE_ERR_ARG; }
#undef GO
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_nhwc_to_nchw(int dst_dtype, int src_dtype, const void* x, int ld, int B, int C, int H, int W, void* y, void* stream) {
    const size_t total = (size_t)B * C * H * W;
    if (total == 0) return MVE_OK;
    MVE_CHECK(x && y && ld >= C, MVE_ERR_ARG, "nhwc_to_nchw: bad arguments");
    const unsigned grid = mve_cdiv(total, NT);
    hipStream_t s = (hipStream_t)stream;
#define GO(SRC, DST) k_nhwc_to_nchw<SRC, DST><<<grid, NT, 0, s>>>((const SRC*)x, ld, B, C, H * W, (DST*)y)
    if (src_dtype == MVE_F32) {
        if (dst_dtype == MVE_F32) GO(float, float);
        else if (dst_dtype == MVE_F16) GO(float, f16);
        else if (dst_dtype == MVE_BF16) GO(float, bf16);
        else { mve_set_error("nhwc_to_nchw: bad dst dtype"); return MVE_ERR_ARG; }
    } else if (src_dtype == MVE_F16) {
        if (dst_dtype == MVE_F32) GO(f16, float);
        else if (dst_dtype == MVE_F16) GO(f16, f16);
        else { mve_set_error("nhwc_to_nchw: bad dst dtype"); return MVE_ERR_ARG; }
    } else if (src_dtype == MVE_BF16) {
        if (dst_dtype == MVE_F32) GO(bf16, float);
        else if (dst_dtype == MVE_BF16) GO(bf16, bf16);
        else { mve_set_error("nhwc_to_nchw: bad dst dtype"); return MVE_ERR_ARG; }
    } else { mve_set_error("nhwc_to_nchw: bad src dtype"); return MVE_ERR_ARG; }
#undef GO
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_timestep_embedding(int dtype, const float* t, int B, int dim, void* out, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(t && out && dim > 0 && dim % 2 == 0, MVE_ERR_ARG, "timestep_embedding: bad arguments");
    const unsigned grid = mve_cdiv((size_t)B * dim, 256);
    if (dtype == MVE_F16) k_timestep_embedding<F16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(t, B, dim, (f16*)out);
    else if (dtype == MVE_BF16) k_timestep_embedding<BF16Tag><<<grid, 256, 0, (hipStream_t)stream>>>(t, B, dim, (bf16*)out);
    else { mve_set_error("timestep_embedding: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_silu(int dtype, const void* x, void* y, size_t n, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(x && y && n % 8 == 0, MVE_ERR_ARG, "silu: n must be a multiple of 8");
    const unsigned grid = mve_cdiv(n / 8, NT);
    if (dtype == MVE_F16) k_silu<F16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const f16*)x, (f16*)y, n / 8);
    else if (dtype == MVE_BF16) k_silu<BF16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const bf16*)x, (bf16*)y, n / 8);
    else { mve_set_error("silu: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_axpy(int dtype, const void* a, const void* b, float alpha, void* y, size_t n, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(a && b && y && n % 8 == 0, MVE_ERR_ARG, "axpy: n must be a multiple of 8");
    const unsigned grid = mve_cdiv(n / 8, NT);
    if (dtype == MVE_F16) k_axpy<F16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const f16*)a, (const f16*)b, alpha, (f16*)y, n / 8);
    else if (dtype == MVE_BF16) k_axpy<BF16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const bf16*)a, (const bf16*)b, alpha, (bf16*)y, n / 8);
    else { mve_set_error("axpy: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_axpy_pair(int dtype, const void* a, const void* a_lo, const void* b, float alpha, void* y, void* y_lo, size_t n, void* stream) {
    if (!y_lo) {
        MVE_CHECK(!a_lo, MVE_ERR_ARG, "axpy_pair: a low half needs somewhere to go (y_lo)");
        return mve_axpy(dtype, a, b, alpha, y, n, stream);
    }
    if (n == 0) return MVE_OK;
    MVE_CHECK(a && b && y && n % 8 == 0, MVE_ERR_ARG, "axpy_pair: n must be a multiple of 8");
    const unsigned grid = mve_cdiv(n / 8, NT);
    if (dtype == MVE_F16) k_axpy_pair<F16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const f16*)a, (const unsigned char*)a_lo, (const f16*)b, alpha, (f16*)y, (unsigned char*)y_lo, n / 8);
    else if (dtype == MVE_BF16) k_axpy_pair<BF16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const bf16*)a, (const unsigned char*)a_lo, (const bf16*)b, alpha, (bf16*)y, (unsigned char*)y_lo, n / 8);
    else { mve_set_error("axpy_pair: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_cfg_combine(const float* uncond, const float* text, float guidance_scale, float* out, size_t n, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(uncond && text && out, MVE_ERR_ARG, "cfg_combine: null pointer");
    k_cfg<<<mve_cdiv(n, NT), NT, 0, (hipStream_t)stream>>>(uncond, text, guidance_scale, out, n);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_x0_prediction(const float* d_latents_scaled, const float* d_noise_pred, float sqrt_alpha_bar, float sqrt_one_minus_alpha_bar, size_t n,
                      float* d_x0, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(d_latents_scaled && d_noise_pred && d_x0 && sqrt_alpha_bar > 0.0f, MVE_ERR_ARG, "x0_prediction: bad arguments");
    k_x0_prediction<<<mve_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(d_latents_scaled, d_noise_pred, sqrt_alpha_bar, sqrt_one_minus_alpha_bar, n, d_x0);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_softmax_rows(int dtype, const float* d_scores, size_t lds, int M, int N, void* d_probs, size_t ldp, void* stream) {
    if (M == 0) return MVE_OK;
    MVE_CHECK(d_scores && d_probs && N > 0 && N % 4 == 0 && lds % 4 == 0 && ldp % 4 == 0 && lds >= (size_t)N && ldp >= (size_t)N, MVE_ERR_ARG,
              "softmax_rows: N, lds, ldp must be multiples of 4 (N=%d)", N);
    if (dtype == MVE_F16) k_softmax_rows<F16Tag><<<M, 256, 0, (hipStream_t)stream>>>(d_scores, lds, N, (f16*)d_probs, ldp);
    else if (dtype == MVE_BF16) k_softmax_rows<BF16Tag><<<M, 256, 0, (hipStream_t)stream>>>(d_scores, lds, N, (bf16*)d_probs, ldp);
    else { mve_set_error("softmax_rows: unsupported dtype %d", dtype); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_prelu(int dtype, const void* x, const float* slope, int C, void* y, size_t n, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(x && y && slope && C > 0 && C % 8 == 0 && n % (size_t)C == 0, MVE_ERR_ARG, "prelu: C must be a multiple of 8 dividing n");
    const unsigned grid = mve_cdiv(n / 8, NT);
    if (dtype == MVE_F16) k_prelu<F16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const f16*)x, slope, C / 8, (f16*)y, n / 8);
    else if (dtype == MVE_BF16) k_prelu<BF16Tag><<<grid, NT, 0, (hipStream_t)stream>>>((const bf16*)x, slope, C / 8, (bf16*)y, n / 8);
    else { mve_set_error("prelu: bad dtype"); return MVE_ERR_ARG; }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_pixel_shuffle_add(int io_dtype, const float* d_src, int ld, const void* d_base, int B, int C, int H, int W, int r, void* d_out,
                          void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_src && d_base && d_out && C > 0 && r > 0 && ld >= C * r * r, MVE_ERR_ARG, "pixel_shuffle_add: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (io_dtype == MVE_F32) return pixel_shuffle_add_dst<float>(io_dtype, d_src, ld, d_base, B, C, H, W, r, (float*)d_out, s);
    if (io_dtype == MVE_F16) return pixel_shuffle_add_dst<f16>(io_dtype, d_src, ld, d_base, B, C, H, W, r, (f16*)d_out, s);
    if (io_dtype == MVE_BF16) return pixel_shuffle_add_dst<bf16>(io_dtype, d_src, ld, d_base, B, C, H, W, r, (bf16*)d_out, s);
    mve_set_error("pixel_shuffle_add: bad dtype");
    return MVE_ERR_ARG;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Texture-seam dilation (lib/ops/edge_dilation.py:5-47 of the reference).  One launch per iteration:
//   mask_out = maxpool_k(mask);  every pixel with mask_out - mask > 0.5 copies the image value of the valid
//   pixel of its k x k window that maximises  mask * (d_max - d + 1),  d = Euclidean offset length, first
//   maximum in row-major window order on ties (torch.argmax over the F.unfold axis).
// The reference materialises the 49-fold unfolded mask (and gathers through it); here each pixel scans its
// window in registers: 2 reads + 1 write of the image per iteration instead of ~50.
// ---------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void k_edge_dilate(const float* __restrict__ img, const float* __restrict__ mask, int n, int c,
                                                     int h, int w, int r, float* __restrict__ img_out, float* __restrict__ mask_out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t hw = (size_t)h * w;
    if (i >= (size_t)n * hw) return;
    const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / hw);
    const float* mk = mask + (size_t)b * hw;
    const float m0 = mk[(size_t)y * w + x];
    float mmax = -INFINITY;            // F.max_pool2d pads with -inf
    float best = -INFINITY;
    int by = y, bx = x;
    const float dmax = sqrtf((float)(2 * r * r));
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int yy = y + dy, xx = x + dx;
            const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
            const float mv = in ? mk[(size_t)yy * w + xx] : 0.0f;          // F.unfold pads with zeros
            if (in) mmax = fmaxf(mmax, mv);
            const float score = mv * (dmax - sqrtf((float)(dx * dx + dy * dy)) + 1.0f);
            if (score > best) { best = score; by = yy; bx = xx; }       // strict >: first maximum wins
        }
    const bool fill = (mmax - m0) > 0.5f;
    mask_out[i] = mmax;
    // an all-invalid window (best == 0 at the first, out-of-image-or-zero entry) cannot be a fill pixel, so by/bx are in range
    for (int ch = 0; ch < c; ++ch) {
        const float* src = img + ((size_t)b * c + ch) * hw;
        img_out[((size_t)b * c + ch) * hw + (size_t)y * w + x] = fill ? src[(size_t)by * w + bx] : src[(size_t)y * w + x];
    }
}

}  // namespace

extern "C" int mve_edge_dilation(const float* d_img, const float* d_mask, int n, int c, int h, int w, float radius, int iters,
                                 float* d_img_out, float* d_mask_out, float* d_img_tmp, float* d_mask_tmp, void* stream) {
    MVE_CHECK(d_img && d_mask && d_img_out && d_mask_out, MVE_ERR_ARG, "edge_dilation: null pointer");
    const size_t npix = (size_t)n * h * w;
    hipStream_t s = (hipStream_t)stream;
    const int r = (int)lrintf(radius);          // Python round(): half to even, as lrintf in the default rounding mode
    if (npix == 0) return MVE_OK;
    if (r == 0 || iters <= 0) {
        MVE_HIP(hipMemcpyAsync(d_img_out, d_img, npix * c * sizeof(float), hipMemcpyDeviceToDevice, s));
        MVE_HIP(hipMemcpyAsync(d_mask_out, d_mask, npix * sizeof(float), hipMemcpyDeviceToDevice, s));
        return MVE_OK;
    }
    MVE_CHECK(iters == 1 || (d_img_tmp && d_mask_tmp), MVE_ERR_ARG, "edge_dilation: ping-pong buffers required for iters > 1");
    const float* src_i = d_img;
    const float* src_m = d_mask;
    for (int it = 0; it < iters; ++it) {
        // ping-pong so that the LAST iteration lands in the caller's output buffers
        const bool to_out = ((iters - 1 - it) % 2) == 0;
        float* dst_i = to_out ? d_img_out : d_img_tmp;
        float* dst_m = to_out ? d_mask_out : d_mask_tmp;
        k_edge_dilate<<<mve_cdiv(npix, 256), 256, 0, s>>>(src_i, src_m, n, c, h, w, r, dst_i, dst_m);
        MVE_LAUNCH_CHECK();
        src_i = dst_i;
        src_m = dst_m;
    }
    return MVE_OK;
}
