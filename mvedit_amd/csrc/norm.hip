// GroupNorm(+SiLU) and LayerNorm over NHWC / token-major activations (gfx950).  HBM-bound kernels:
// every byte is read with 16-byte lane loads along the contiguous channel axis, a wave reads 1 KiB
// contiguous per instruction.
//
// GroupNorm (torch.nn.GroupNorm semantics: biased variance over (C/G, H, W), eps inside the sqrt) is
// split in three launches so that the reduction is deterministic (no float atomics):
//   1. gn_partial : grid (split, colgroup, B); each thread owns one 8-channel chunk and walks rows,
//                   block-reduces to per-channel (sum, sumsq) partials in fp32;
//   2. gn_finalize: one thread per (b, group) merges partials in fp64 -> (mean, rstd);
//   3. gn_apply   : y = act((x-mean)*rstd*gamma + beta), optional SiLU, optional second input tensor
//                   (the skip connection of an up block: the concatenated, normalised tensor is
//                   written once, so the following conv reads a single dense input).
// They replace GroupNorm + SiLU inside diffusers' ResnetBlock2D / Transformer2DModel as driven by
// lib/models/architecture/diffusers.py:86-97,139-156 of the reference.
#include "common.h"

namespace {

// A block is 256 threads: TX chunks (of 8 channels) side by side x TY = 256 / TX rows in flight.  TX = 64 for the UNet's widths
// (C >= 320); the VAE's 128 / 256-channel tensors at image resolution use TX = 16 / 32 so that no lane idles.
constexpr int GN_THREADS = 256;

template <class Tag>
__device__ __forceinline__ void load8(const typename Tag::T* p, float (&v)[8]) {
    const typename Tag::V8 x = *reinterpret_cast<const typename Tag::V8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = Tag::to_f32(x[e]);
}

struct GNSrc {
    const void* x1; const void* x2;
    int C1, C2;       // channels of each source (C2 may be 0)
    int HW, B;
};

// chunk index (over the concatenated channel axis) -> source pointer for row r of image b
template <class Tag>
__device__ __forceinline__ const typename Tag::T* gn_chunk_ptr(const GNSrc& s, int b, int r, int chunk) {
    typedef typename Tag::T T;
    const int ch = chunk * 8;
    if (ch < s.C1) return reinterpret_cast<const T*>(s.x1) + ((size_t)b * s.HW + r) * s.C1 + ch;
    return reinterpret_cast<const T*>(s.x2) + ((size_t)b * s.HW + r) * s.C2 + (ch - s.C1);
}

template <class Tag, int GN_TX>
__global__ __launch_bounds__(GN_THREADS) void k_gn_partial(GNSrc s, int nsplit, float* __restrict__ partial) {
    constexpr int GN_TY = GN_THREADS / GN_TX;
    // partial: [B][nsplit][C][2]
    __shared__ float red[GN_TY][GN_TX][16];
    const int C = s.C1 + s.C2;
    const int chunk = blockIdx.y * GN_TX + threadIdx.x;
    const int b = blockIdx.z, split = blockIdx.x;
    const int rows_per = (s.HW + nsplit - 1) / nsplit;
    const int r0 = split * rows_per, r1 = min(s.HW, r0 + rows_per);
    float sum[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sum[e] = 0.f; sq[e] = 0.f; }
    const bool active = chunk * 8 < C;
    if (active) {
        for (int r = r0 + threadIdx.y; r < r1; r += GN_TY) {
            float v[8];
            load8<Tag>(gn_chunk_ptr<Tag>(s, b, r, chunk), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { sum[e] += v[e]; sq[e] += v[e] * v[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.y][threadIdx.x][e] = sum[e]; red[threadIdx.y][threadIdx.x][8 + e] = sq[e]; }
    __syncthreads();
    if (threadIdx.y == 0 && active) {
        float* out = partial + (((size_t)b * nsplit + split) * C + chunk * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int y = 0; y < GN_TY; ++y) { a += red[y][threadIdx.x][e]; q += red[y][threadIdx.x][8 + e]; }
            out[2 * e] = a;
            out[2 * e + 1] = q;
        }
    }
}

__global__ void k_gn_finalize(const float* __restrict__ partial, int B, int nsplit, int C, int G, int HW, float eps,
                              float* __restrict__ stats /* [B][G][2] mean, rstd */) {
    // one wave per (b, group): lanes stride over the nsplit*cpg partials, fixed-order xor-tree merge in fp64
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= B * G) return;
    const int lane = threadIdx.x & 63;
    const int b = i / G, g = i - b * G, cpg = C / G;
    double a = 0.0, q = 0.0;
    for (int t = lane; t < nsplit * cpg; t += 64) {
        const int sp = t / cpg, c = t - sp * cpg;
        const float* p = partial + (((size_t)b * nsplit + sp) * C + (size_t)g * cpg + c) * 2;
        a += (double)p[0];
        q += (double)p[1];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); q += __shfl_xor(q, d, 64); }
    if (lane != 0) return;
    const double n = (double)cpg * HW;
    const double mean = a / n;
    double var = q / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

template <class Tag, int GN_TX>
__global__ __launch_bounds__(GN_THREADS) void k_gn_apply(GNSrc s, int nsplit, int G, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int silu, void* __restrict__ out) {
    constexpr int GN_TY = GN_THREADS / GN_TX;
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    const int C = s.C1 + s.C2;
    const int chunk = blockIdx.y * GN_TX + threadIdx.x;
    if (chunk * 8 >= C) return;
    const int b = blockIdx.z, split = blockIdx.x;
    const int rows_per = (s.HW + nsplit - 1) / nsplit;
    const int r0 = split * rows_per, r1 = min(s.HW, r0 + rows_per);
    const int cpg = C / G;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 8 + e;
        const int g = c / cpg;
        const float mean = stats[2 * (b * G + g)], rstd = stats[2 * (b * G + g) + 1];
        sc[e] = rstd * gamma[c];
        sh[e] = beta[c] - mean * sc[e];
    }
    T* o = reinterpret_cast<T*>(out);
    for (int r = r0 + threadIdx.y; r < r1; r += GN_TY) {
        float v[8];
        load8<Tag>(gn_chunk_ptr<Tag>(s, b, r, chunk), v);
        V8 pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = v[e] * sc[e] + sh[e];
            if (silu) y = y / (1.0f + __expf(-y));
            pk[e] = Tag::from_f32(y);
        }
        *reinterpret_cast<V8*>(o + ((size_t)b * s.HW + r) * C + chunk * 8) = pk;
    }
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over the last axis (C <= 2048, C % 8 == 0): one wave per row, two passes over registers.
// ---------------------------------------------------------------------------------------------------
template <class Tag, int MAXC8>   // MAXC8: chunks per lane
__global__ __launch_bounds__(256) void k_layernorm(const void* __restrict__ x, int ldx, void* __restrict__ y, int ldy, int M,
                                                   int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps) {
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = reinterpret_cast<const T*>(x) + (size_t)row * ldx;
    const int nchunk = C / 8;
    float v[MAXC8][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC8; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            load8<Tag>(xr + c * 8, v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC8; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) q += __shfl_xor(q, d, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    T* yr = reinterpret_cast<T*>(y) + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXC8; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            V8 pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = c * 8 + e;
                pk[e] = Tag::from_f32((v[i][e] - mean) * rstd * gamma[ch] + beta[ch]);
            }
            *reinterpret_cast<V8*>(yr + c * 8) = pk;
        }
    }
}

int gn_tx(int C) { return C <= 128 ? 16 : (C <= 256 ? 32 : 64); }

int gn_nsplit(int B, int HW, int C) {
    if (gn_tx(C) < 64) {      // narrow tensors are the large-image ones: ~512 rows per block, independent of the batch
        const int ns = HW / 512;
        return ns < 1 ? 1 : (ns > 512 ? 512 : ns);
    }
    const int ncolgroups = (C / 8 + 63) / 64;
    // enough blocks to fill 256 CUs several times over, but at least 32 rows per block
    int want = (256 * 8 + B * ncolgroups - 1) / (B * ncolgroups);
    int maxsplit = HW / 32 > 0 ? HW / 32 : 1;
    int ns = want < maxsplit ? want : maxsplit;
    return ns < 1 ? 1 : (ns > 64 ? 64 : ns);
}

template <class Tag>
int gn_run(const GNSrc& s, int G, float eps, const float* gamma, const float* beta, int silu, void* out, float* ws,
           hipStream_t st) {
    const int C = s.C1 + s.C2;
    const int tx = gn_tx(C);
    const int ncg = (C / 8 + tx - 1) / tx;
    const int ns = gn_nsplit(s.B, s.HW, C);
    float* partial = ws;
    float* stats = ws + (size_t)s.B * ns * C * 2;
    dim3 grid(ns, ncg, s.B), block(tx, GN_THREADS / tx);
    if (tx == 16) k_gn_partial<Tag, 16><<<grid, block, 0, st>>>(s, ns, partial);
    else if (tx == 32) k_gn_partial<Tag, 32><<<grid, block, 0, st>>>(s, ns, partial);
    else k_gn_partial<Tag, 64><<<grid, block, 0, st>>>(s, ns, partial);
    MVE_LAUNCH_CHECK();
    k_gn_finalize<<<mve_cdiv(s.B * G, 4), 256, 0, st>>>(partial, s.B, ns, C, G, s.HW, eps, stats);
    MVE_LAUNCH_CHECK();
    if (tx == 16) k_gn_apply<Tag, 16><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
    else if (tx == 32) k_gn_apply<Tag, 32><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
    else k_gn_apply<Tag, 64><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

template <class Tag>
int ln_run(const void* x, int ldx, void* y, int ldy, int M, int C, const float* gamma, const float* beta, float eps,
           hipStream_t st) {
    const int per_lane = (C / 8 + 63) / 64;
    const unsigned grid = mve_cdiv(M, 4);
    if (per_lane <= 1) k_layernorm<Tag, 1><<<grid, 256, 0, st>>>(x, ldx, y, ldy, M, C, gamma, beta, eps);
    else if (per_lane == 2) k_layernorm<Tag, 2><<<grid, 256, 0, st>>>(x, ldx, y, ldy, M, C, gamma, beta, eps);
    else if (per_lane == 3) k_layernorm<Tag, 3><<<grid, 256, 0, st>>>(x, ldx, y, ldy, M, C, gamma, beta, eps);
    else k_layernorm<Tag, 4><<<grid, 256, 0, st>>>(x, ldx, y, ldy, M, C, gamma, beta, eps);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // namespace

extern "C" {

size_t mve_groupnorm_workspace_bytes(int B, int HW, int C, int G) {
    if (B <= 0 || HW <= 0 || C <= 0) return 64;
    const int ns = gn_nsplit(B, HW, C);
    return sizeof(float) * ((size_t)B * ns * C * 2 + (size_t)B * G * 2 + 16);
}

int mve_groupnorm_silu(int dtype, const void* x1, int C1, const void* x2, int C2, int B, int HW, int G, float eps,
                       const float* gamma, const float* beta, int silu, void* out, void* workspace, void* stream) {
    if (B == 0 || HW == 0) return MVE_OK;
    const int C = C1 + C2;
    MVE_CHECK(x1 && out && gamma && beta && workspace && (C2 == 0 || x2), MVE_ERR_ARG, "groupnorm: null pointer");
    MVE_CHECK(C1 % 8 == 0 && C2 % 8 == 0 && G > 0 && C % G == 0, MVE_ERR_ARG,
              "groupnorm: C1=%d C2=%d must be multiples of 8 and C divisible by G=%d", C1, C2, G);
    GNSrc s;
    s.x1 = x1; s.x2 = x2; s.C1 = C1; s.C2 = C2; s.HW = HW; s.B = B;
    if (dtype == MVE_F16) return gn_run<F16Tag>(s, G, eps, gamma, beta, silu, out, (float*)workspace, (hipStream_t)stream);
    if (dtype == MVE_BF16) return gn_run<BF16Tag>(s, G, eps, gamma, beta, silu, out, (float*)workspace, (hipStream_t)stream);
    mve_set_error("groupnorm: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

int mve_layernorm(int dtype, const void* x, int ldx, void* y, int ldy, int M, int C, const float* gamma, const float* beta,
                  float eps, void* stream) {
    if (M == 0) return MVE_OK;
    MVE_CHECK(x && y && gamma && beta, MVE_ERR_ARG, "layernorm: null pointer");
    MVE_CHECK(C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && ldy % 8 == 0, MVE_ERR_ARG,
              "layernorm: C=%d must be a multiple of 8 and <= 2048", C);
    if (dtype == MVE_F16) return ln_run<F16Tag>(x, ldx, y, ldy, M, C, gamma, beta, eps, (hipStream_t)stream);
    if (dtype == MVE_BF16) return ln_run<BF16Tag>(x, ldx, y, ldy, M, C, gamma, beta, eps, (hipStream_t)stream);
    mve_set_error("layernorm: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

}  // extern "C"
