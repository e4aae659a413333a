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
#include "ln_core.h"

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
    const void* x1_lo; const void* x2_lo;      // residual_pair mode: 8-bit low halves of the sources (lo8, common.h: the value is hi + lo), or null
};

// chunk index (over the concatenated channel axis) -> source pointer for row r of image b
template <class Tag>
__device__ __forceinline__ const typename Tag::T* gn_chunk_ptr(const GNSrc& s, int b, int r, int chunk) {
    typedef typename Tag::T T;
    const int ch = chunk * 8;
    if (ch < s.C1) return reinterpret_cast<const T*>(s.x1) + ((size_t)b * s.HW + r) * s.C1 + ch;
    return reinterpret_cast<const T*>(s.x2) + ((size_t)b * s.HW + r) * s.C2 + (ch - s.C1);
}

// 8 channels of row r as floats; PAIR: hi + lo of the stream pair (exact in fp32)
template <class Tag, bool PAIR>
__device__ __forceinline__ void gn_load8(const GNSrc& s, int b, int r, int chunk, float (&v)[8]) {
    typedef typename Tag::T T;
    const T* ph = gn_chunk_ptr<Tag>(s, b, r, chunk);
    load8<Tag>(ph, v);
    if constexpr (PAIR) {
        const int ch = chunk * 8;
        const T* base_h = reinterpret_cast<const T*>(ch < s.C1 ? s.x1 : s.x2);
        const unsigned char* base_l = reinterpret_cast<const unsigned char*>(ch < s.C1 ? s.x1_lo : s.x2_lo);
        if (base_l) {                      // lo8: one byte per element at the element offset of the high half
            const u32x2 l = *reinterpret_cast<const u32x2*>(base_l + (ph - base_h));
            float a[4], b[4];
            mve_lo8_unpack4(l[0], a);
            mve_lo8_unpack4(l[1], b);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
        }
    }
}

template <class Tag, int GN_TX, bool PAIR = false>
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
        // four rows requested before the first one is summed (round 4: one 16-byte load in flight per thread left the pass at 3.9 TB/s); the
        // per-thread summation order over rows is unchanged: identical bits
        int r = r0 + threadIdx.y;
        for (; r + 3 * GN_TY < r1; r += 4 * GN_TY) {
            float v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) gn_load8<Tag, PAIR>(s, b, r + k * GN_TY, chunk, v[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) { sum[e] += v[k][e]; sq[e] += v[k][e] * v[k][e]; }
        }
        for (; r < r1; r += GN_TY) {
            float v[8];
            gn_load8<Tag, PAIR>(s, b, r, chunk, v);
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

template <class Tag, int GN_TX, bool PAIR = false>
__global__ __launch_bounds__(GN_THREADS) void k_gn_apply(GNSrc s, int nsplit, int G, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int silu, void* __restrict__ out) {
    constexpr int GN_TY = GN_THREADS / GN_TX;
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    const int C = s.C1 + s.C2;
    const int chunk = blockIdx.y * GN_TX + threadIdx.x;
    if (chunk * 8 >= C) return;
    // (round 5) the apply pass walks the tensor in the REVERSE order of the statistics pass: what that pass read last is read first here, while it
    // is still in the Infinity Cache (3-7 % on the level-0 tensors, tools/gn_pair_bench.py; pure index remap: same bits)
    const int b = (int)gridDim.z - 1 - (int)blockIdx.z, split = (int)gridDim.x - 1 - (int)blockIdx.x;
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
    auto finish = [&](const float (&v)[8], int r) {
        V8 pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = v[e] * sc[e] + sh[e];
            if (silu) y = y / (1.0f + __expf(-y));
            pk[e] = Tag::from_f32(y);
        }
        *reinterpret_cast<V8*>(o + ((size_t)b * s.HW + r) * C + chunk * 8) = pk;
    };
    // four rows loaded before the first store (a store between two loads orders them: the sources may alias `out` as far as hipcc knows)
    int r = r0 + threadIdx.y;
    for (; r + 3 * GN_TY < r1; r += 4 * GN_TY) {
        float v[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) gn_load8<Tag, PAIR>(s, b, r + k * GN_TY, chunk, v[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) finish(v[k], r + k * GN_TY);
    }
    for (; r < r1; r += GN_TY) {
        float v[8];
        gn_load8<Tag, PAIR>(s, b, r, chunk, v);
        finish(v, r);
    }
}

// ---------------------------------------------------------------------------------------------------
// One-launch GroupNorm for tensors whose (image, channel chunk) slice fits the registers of one block: a chunk is lcm(C / G, 8) channels
// (whole groups AND whole 16-byte vectors: 40 channels for the UNet's 320 / 640 / 1280-wide tensors, 120 for 960 / 1920, 80 for 2560),
// the block loads all HW rows of it once (<= 20 vectors per thread, kept packed), takes mean and centred variance per group out of the
// registers (two passes, fp32, fixed-order tree), and writes the normalised (+SiLU) slice: one read + one write of the tensor and one
// launch instead of 2 reads + 1 write and three.  Which path a tensor takes depends on (HW, C, G) only -- never on the batch.
// ---------------------------------------------------------------------------------------------------
constexpr int GNF_MAX_GROUPS = 4;

template <int NT>
__device__ __forceinline__ void gnf_block_sum(float (&v)[GNF_MAX_GROUPS], float* red /* [NT / 64][GNF_MAX_GROUPS] */) {
#pragma unroll
    for (int g = 0; g < GNF_MAX_GROUPS; ++g) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[g] += __shfl_xor(v[g], d, 64);
    }
    __syncthreads();                                       // the previous round's readers are done with `red`
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int g = 0; g < GNF_MAX_GROUPS; ++g) red[(threadIdx.x >> 6) * GNF_MAX_GROUPS + g] = v[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GNF_MAX_GROUPS; ++g) {
        float a = 0.f;
        for (int w = 0; w < NT / 64; ++w) a += red[w * GNF_MAX_GROUPS + g];
        v[g] = a;
    }
}

// PAIR (round 5): the sources are stream pairs (hi + lo8, common.h).  The 16-bit halves stay packed in registers as before; the 8-bit low halves
// are re-read from memory in each of the three passes (8 bytes per vector, L2-resident: the slice was fetched a moment ago) instead of costing
// another 2 registers per vector next to MAXI = 20 of them.  Until round 5 a pair took the three-launch path at every size: 28 of the 61 GroupNorms
// of an SD-1.5 forward, ~20 us each at 8 images per rank.
template <class Tag, int NT, int MAXI, bool PAIR = false>
__global__ __launch_bounds__(NT) void k_gn_fused(GNSrc s, int G, int W8, int nchunk, int xcd_map, float eps, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, int silu, void* __restrict__ out) {
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    __shared__ float red[(NT / 64) * GNF_MAX_GROUPS];
    __shared__ float tab[2][128];                          // per chunk channel: scale, shift
    const int C = s.C1 + s.C2, cpg = C / G, cg = W8 * 8 / cpg;
    int b, chunk;
    if (xcd_map) {                                         // the chunks of an image on ONE XCD (block id % 8), close in time: they share its rows' lines
        const int id = blockIdx.x;
        b = 8 * (id / (8 * nchunk)) + (id & 7);
        chunk = (id >> 3) % nchunk;
    } else {
        b = blockIdx.x / nchunk;
        chunk = blockIdx.x - b * nchunk;
    }
    const int total = s.HW * W8;
    V8 raw[MAXI];
    // the 8 values of vector `it` (row, c8): the packed 16-bit half, plus its low half when the source is a pair
    auto vals = [&](int it, int row, int c8, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Tag::to_f32(raw[it][e]);
        if constexpr (PAIR) {
            const int ch = (chunk * W8 + c8) * 8;
            const bool first = ch < s.C1;
            const unsigned char* base = reinterpret_cast<const unsigned char*>(first ? s.x1_lo : s.x2_lo);
            if (base) {
                const u32x2 l = *reinterpret_cast<const u32x2*>(base + ((size_t)b * s.HW + row) * (first ? s.C1 : s.C2) + (first ? ch : ch - s.C1));
                float a4[4], b4[4];
                mve_lo8_unpack4(l[0], a4);
                mve_lo8_unpack4(l[1], b4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += a4[e]; v[4 + e] += b4[e]; }
            }
        }
    };
    float acc[GNF_MAX_GROUPS] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int i = threadIdx.x + it * NT;
        if (i < total) {
            const int row = i / W8, c8 = i - row * W8;
            raw[it] = *reinterpret_cast<const V8*>(gn_chunk_ptr<Tag>(s, b, row, chunk * W8 + c8));
            const int g0 = (c8 * 8) / cpg, eb = (g0 + 1) * cpg - c8 * 8;
            float lo = 0.f, hi = 0.f;
            float x8[8];
            vals(it, row, c8, x8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = x8[e]; if (e < eb) lo += v; else hi += v; }
#pragma unroll
            for (int g = 0; g < GNF_MAX_GROUPS; ++g) acc[g] += (g == g0 ? lo : 0.f) + (g == g0 + 1 ? hi : 0.f);
        }
    }
    gnf_block_sum<NT>(acc, red);
    const float inv_n = 1.0f / ((float)cpg * (float)s.HW);
    float mean[GNF_MAX_GROUPS];
#pragma unroll
    for (int g = 0; g < GNF_MAX_GROUPS; ++g) { mean[g] = acc[g] * inv_n; acc[g] = 0.f; }
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int i = threadIdx.x + it * NT;
        if (i < total) {
            const int row = i / W8, c8 = i - row * W8;
            const int g0 = (c8 * 8) / cpg, eb = (g0 + 1) * cpg - c8 * 8;
            float m0 = 0.f, m1 = 0.f;
#pragma unroll
            for (int g = 0; g < GNF_MAX_GROUPS; ++g) { m0 = g == g0 ? mean[g] : m0; m1 = g == g0 + 1 ? mean[g] : m1; }
            float lo = 0.f, hi = 0.f;
            float x8[8];
            vals(it, row, c8, x8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = x8[e];
                if (e < eb) { const float d = v - m0; lo = __builtin_fmaf(d, d, lo); } else { const float d = v - m1; hi = __builtin_fmaf(d, d, hi); }
            }
#pragma unroll
            for (int g = 0; g < GNF_MAX_GROUPS; ++g) acc[g] += (g == g0 ? lo : 0.f) + (g == g0 + 1 ? hi : 0.f);
        }
    }
    gnf_block_sum<NT>(acc, red);
    if ((int)threadIdx.x < W8 * 8) {
        const int g = threadIdx.x / cpg, c = chunk * W8 * 8 + threadIdx.x;
        float var = 0.f, m = 0.f;
#pragma unroll
        for (int k = 0; k < GNF_MAX_GROUPS; ++k) { var = k == g ? acc[k] : var; m = k == g ? mean[k] : m; }
        const float sc = rsqrtf(var * inv_n + eps) * gamma[c];
        tab[0][threadIdx.x] = sc;
        tab[1][threadIdx.x] = beta[c] - m * sc;
    }
    __syncthreads();
    (void)cg;
    T* o = reinterpret_cast<T*>(out);
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int i = threadIdx.x + it * NT;
        if (i < total) {
            const int row = i / W8, c8 = i - row * W8;
            V8 pk;
            float x8[8];
            vals(it, row, c8, x8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = __builtin_fmaf(x8[e], tab[0][c8 * 8 + e], tab[1][c8 * 8 + e]);
                if (silu) y = y / (1.0f + __expf(-y));
                pk[e] = Tag::from_f32(y);
            }
            *reinterpret_cast<V8*>(o + ((size_t)b * s.HW + row) * C + (size_t)(chunk * W8 + c8) * 8) = pk;
        }
    }
}

int g_gn_fused_max_hw = -1;          // -1: read MVE_GN_FUSED_MAX_HW (default 1024); 0 disables the one-launch path
int g_gn_fused_pair_max_hw = -1;     // -1: read MVE_GN_FUSED_PAIR_MAX_HW (default 256): stream PAIRS take the one-launch path up to this many pixels per image
int gn_fused_pair_max_hw() {
    if (g_gn_fused_pair_max_hw < 0) {
        const char* e = getenv("MVE_GN_FUSED_PAIR_MAX_HW");
        g_gn_fused_pair_max_hw = e ? atoi(e) : 256;
    }
    return g_gn_fused_pair_max_hw;
}

// -> vectors per row of a chunk (0: the tensor does not take the one-launch path), threads and vectors per thread
int gnf_plan(int HW, int C, int G, int* nt, int* maxi) {
    if (g_gn_fused_max_hw < 0) {
        const char* e = getenv("MVE_GN_FUSED_MAX_HW");
        g_gn_fused_max_hw = e ? atoi(e) : 1024;
    }
    if (HW > g_gn_fused_max_hw || G <= 0 || C % G) return 0;
    const int cpg = C / G;
    int ch = cpg;
    while (ch % 8) ch += cpg;                              // lcm(cpg, 8)
    if (cpg < 8 || ch > 128 || C % ch || ch / cpg > GNF_MAX_GROUPS) return 0;
    const int W8 = ch / 8;
    const long long total = (long long)HW * W8;
    if (total <= 256 * 20) *nt = 256;
    else if (total <= 1024 * 20) *nt = 1024;
    else return 0;
    const int per = (int)((total + *nt - 1) / *nt);
    *maxi = per <= 4 ? 4 : (per <= 8 ? 8 : (per <= 12 ? 12 : (per <= 16 ? 16 : 20)));
    return W8;
}

template <class Tag, int NT, bool PAIR = false>
int gnf_launch(const GNSrc& s, int G, int W8, int maxi, float eps, const float* gamma, const float* beta, int silu, void* out, hipStream_t st) {
    const int nchunk = (s.C1 + s.C2) / (W8 * 8);
    const int xcd = (s.B % 8 == 0) ? 1 : 0;
    const unsigned grid = (unsigned)(s.B * nchunk);
    switch (maxi) {
        case 4: k_gn_fused<Tag, NT, 4, PAIR><<<grid, NT, 0, st>>>(s, G, W8, nchunk, xcd, eps, gamma, beta, silu, out); break;
        case 8: k_gn_fused<Tag, NT, 8, PAIR><<<grid, NT, 0, st>>>(s, G, W8, nchunk, xcd, eps, gamma, beta, silu, out); break;
        case 12: k_gn_fused<Tag, NT, 12, PAIR><<<grid, NT, 0, st>>>(s, G, W8, nchunk, xcd, eps, gamma, beta, silu, out); break;
        case 16: k_gn_fused<Tag, NT, 16, PAIR><<<grid, NT, 0, st>>>(s, G, W8, nchunk, xcd, eps, gamma, beta, silu, out); break;
        default: k_gn_fused<Tag, NT, 20, PAIR><<<grid, NT, 0, st>>>(s, G, W8, nchunk, xcd, eps, gamma, beta, silu, out); break;
    }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over the last axis (C <= 2048, C % 8 == 0): one wave per row, two passes over registers.
// ---------------------------------------------------------------------------------------------------
// R rows per wave: a 320-channel row is 640 bytes -- 40 of the 64 lanes load 16 bytes each -- and with one row per wave a CU has ~20 KiB of loads
// in flight, a quarter of what 5 TB/s needs at HBM latency (round 3 measured 2.9 TB/s on the 64 x 64 level).  R = 4 (C <= 512) / 2 (C <= 1024)
// rows are requested before the first one is reduced.  A row's arithmetic (lanes, order, shuffle tree) does not change: identical bits.
template <class Tag, int MAXC8, int R>   // MAXC8: chunks per lane
__global__ __launch_bounds__(256) void k_layernorm(const void* __restrict__ x, int ldx, void* __restrict__ y, int ldy, int M,
                                                   int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, const void* __restrict__ x_lo) {
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= M) return;
    const int nchunk = C / 8;
    V8 raw[R][MAXC8];
    u32x2 rawl[R][MAXC8];                 // residual_pair mode: the 8-bit low halves (lo8, common.h)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r < M ? row0 + r : M - 1;
#pragma unroll
        for (int i = 0; i < MAXC8; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) {
                raw[r][i] = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(x) + (size_t)row * ldx + c * 8);
                if (x_lo) rawl[r][i] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(x_lo) + (size_t)row * ldx + c * 8);   // residual_pair mode (uniform branch)
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        float v[MAXC8][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC8; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = Tag::to_f32(raw[r][i][e]);
                if (x_lo) {
                    float a[4], b[4];
                    mve_lo8_unpack4(rawl[r][i][0], a);
                    mve_lo8_unpack4(rawl[r][i][1], b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[i][e] += a[e]; v[i][4 + e] += b[e]; }
                }
                s = mve_ln_sum8(v[i], s);
            }
        }
        // (the row arithmetic lives in ln_core.h: the GEMM epilogue that normalises its own output rows shares it, bit for bit)
        s = mve_ln_wave_sum(s);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC8; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) q = mve_ln_sq8(v[i], mean, q);
        }
        q = mve_ln_wave_sum(q);
        const float rstd = rsqrtf(q / (float)C + eps);
        T* yr = reinterpret_cast<T*>(y) + (size_t)row * ldy;
#pragma unroll
        for (int i = 0; i < MAXC8; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) *reinterpret_cast<V8*>(yr + c * 8) = mve_ln_out8<Tag>(v[i], mean, rstd, gamma + c * 8, beta + c * 8);
        }
    }
}

// Lanes along the channel chunks of a row.  C <= 256 (the VAE's image-resolution tensors): 16 / 32.  Above that (round 5) a width that DIVIDES the
// number of 8-channel chunks, so that no lane idles: 320 channels = 40 chunks ran on 64-wide blocks with 24 idle lanes, and the pair-reading kernels
// (unpacking + SiLU on 62 % of the lanes) could not hide that behind the memory time -- 2.9 TB/s on the level-0 stream tensors against 4.5 for the plain
// ones.  40-wide blocks (6 rows of 40 lanes; also 80 / 120 / 240 chunks): level-0 C = 320 plain 0.108 -> 0.092 ms, pair 0.229 -> 0.179; C = 640 pair
// 0.439 -> 0.333 (tools/gn_pair_bench.py, 64 images).
int gn_tx(int C) {
    if (C <= 128) return 16;
    if (C <= 256) return 32;
    const int ch = C / 8;
    // (8-wide blocks -- 128-byte row pieces -- lose more to the memory system than idle lanes cost: measured; 40 = the 320 / 960-channel tensors, 6 rows of 40 lanes per block)
    return ch % 64 == 0 ? 64 : (ch % 32 == 0 ? 32 : (ch % 40 == 0 ? 40 : (ch % 16 == 0 ? 16 : 64)));
}

int gn_nsplit(int B, int HW, int C) {
    if (C <= 256) {           // narrow tensors are the large-image ones: ~512 rows per block, independent of the batch
        const int ns = HW / 512;
        return ns < 1 ? 1 : (ns > 512 ? 512 : ns);
    }
    // a function of the rows per image and the channel count only: the grouping of the fp32 partial sums -- and so the statistics' last bits --
    // must not depend on the batch (batch invariance).  16 rows per thread: 64 rows per block of 64-wide blocks, 96 / 128 / 256 for 40 / 32 / 16-wide
    (void)B;
    const int ns = HW / (16 * (GN_THREADS / gn_tx(C)));
    return ns < 1 ? 1 : (ns > 64 ? 64 : ns);
}

template <class Tag>
int gn_run(const GNSrc& s, int G, float eps, const float* gamma, const float* beta, int silu, void* out, float* ws,
           hipStream_t st) {
    const int C = s.C1 + s.C2;
    int nt = 0, maxi = 0;
    const bool pair = s.x1_lo || s.x2_lo;
    if (pair) {
        // (round 5) small tensors take the one-launch kernel as pairs too -- the choice depends on (HW, C, G) only, as for plain tensors
        // (1024-thread blocks with > 8 vectors per thread -- the 960 / 1920-channel concatenations at 32 x 32 -- would spill next to the unpacking: those
        // stay on the three launches)
        // (round 6) ... up to gn_fused_pair_max_hw() = 256 pixels per image (the 16 x 16 and 8 x 8 levels): round 5 measured the pair-reading one-launch
        // kernel at every size <= 1024 and reverted it -- +0.4 ms per 64-image step, all of it on the 32 x 32 tensors (20 packed vectors per thread
        // next to the unpacking), against -0.2 ms at 8 images per rank (profiles/r05_gn_fused_pair_reverted.log).  Below 32 x 32 the slices are 4-8
        // vectors per thread and the three launches are ~5 us of work under ~20 us of launch chain at any batch.
        const int W8 = s.HW <= gn_fused_pair_max_hw() ? gnf_plan(s.HW, C, G, &nt, &maxi) : 0;
        if (W8 && !(nt == 1024 && maxi > 8))
            return nt == 256 ? gnf_launch<Tag, 256, true>(s, G, W8, maxi, eps, gamma, beta, silu, out, st)
                             : gnf_launch<Tag, 1024, true>(s, G, W8, maxi, eps, gamma, beta, silu, out, st);
        const int tx = gn_tx(C);
        const int ncg = (C / 8 + tx - 1) / tx;
        const int ns = gn_nsplit(s.B, s.HW, C);
        float* partial = ws;
        float* stats = ws + (size_t)s.B * ns * C * 2;
        dim3 grid(ns, ncg, s.B), block(tx, GN_THREADS / tx);
        if (tx == 40) k_gn_partial<Tag, 40, true><<<grid, block, 0, st>>>(s, ns, partial);
        else if (tx == 16) k_gn_partial<Tag, 16, true><<<grid, block, 0, st>>>(s, ns, partial);
        else if (tx == 32) k_gn_partial<Tag, 32, true><<<grid, block, 0, st>>>(s, ns, partial);
        else k_gn_partial<Tag, 64, true><<<grid, block, 0, st>>>(s, ns, partial);
        MVE_LAUNCH_CHECK();
        k_gn_finalize<<<mve_cdiv(s.B * G, 4), 256, 0, st>>>(partial, s.B, ns, C, G, s.HW, eps, stats);
        MVE_LAUNCH_CHECK();
        if (tx == 40) k_gn_apply<Tag, 40, true><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
        else if (tx == 16) k_gn_apply<Tag, 16, true><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
        else if (tx == 32) k_gn_apply<Tag, 32, true><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
        else k_gn_apply<Tag, 64, true><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
        MVE_LAUNCH_CHECK();
        return MVE_OK;
    }
    if (const int W8 = gnf_plan(s.HW, C, G, &nt, &maxi))
        return nt == 256 ? gnf_launch<Tag, 256>(s, G, W8, maxi, eps, gamma, beta, silu, out, st)
                         : gnf_launch<Tag, 1024>(s, G, W8, maxi, eps, gamma, beta, silu, out, st);
    const int tx = gn_tx(C);
    const int ncg = (C / 8 + tx - 1) / tx;
    const int ns = gn_nsplit(s.B, s.HW, C);
    float* partial = ws;
    float* stats = ws + (size_t)s.B * ns * C * 2;
    dim3 grid(ns, ncg, s.B), block(tx, GN_THREADS / tx);
    if (tx == 40) k_gn_partial<Tag, 40><<<grid, block, 0, st>>>(s, ns, partial);
    else if (tx == 16) k_gn_partial<Tag, 16><<<grid, block, 0, st>>>(s, ns, partial);
    else if (tx == 32) k_gn_partial<Tag, 32><<<grid, block, 0, st>>>(s, ns, partial);
    else k_gn_partial<Tag, 64><<<grid, block, 0, st>>>(s, ns, partial);
    MVE_LAUNCH_CHECK();
    k_gn_finalize<<<mve_cdiv(s.B * G, 4), 256, 0, st>>>(partial, s.B, ns, C, G, s.HW, eps, stats);
    MVE_LAUNCH_CHECK();
    if (tx == 40) k_gn_apply<Tag, 40><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
    else if (tx == 16) k_gn_apply<Tag, 16><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
    else if (tx == 32) k_gn_apply<Tag, 32><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
    else k_gn_apply<Tag, 64><<<grid, block, 0, st>>>(s, ns, G, stats, gamma, beta, silu, out);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

template <class Tag>
int ln_run(const void* x, int ldx, void* y, int ldy, int M, int C, const float* gamma, const float* beta, float eps,
           hipStream_t st, const void* x_lo) {
    const int per_lane = (C / 8 + 63) / 64;
    if (per_lane <= 1) k_layernorm<Tag, 1, 4><<<mve_cdiv(M, 16), 256, 0, st>>>(x, ldx, y, ldy, M, C, gamma, beta, eps, x_lo);
    else if (per_lane == 2) k_layernorm<Tag, 2, 2><<<mve_cdiv(M, 8), 256, 0, st>>>(x, ldx, y, ldy, M, C, gamma, beta, eps, x_lo);
    else if (per_lane == 3) k_layernorm<Tag, 3, 1><<<mve_cdiv(M, 4), 256, 0, st>>>(x, ldx, y, ldy, M, C, gamma, beta, eps, x_lo);
    else k_layernorm<Tag, 4, 1><<<mve_cdiv(M, 4), 256, 0, st>>>(x, ldx, y, ldy, M, C, gamma, beta, eps, x_lo);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // namespace

extern "C" {

int mve_groupnorm_tune(int fused_max_hw) {
    int nt, maxi;
    (void)gnf_plan(1, 8, 1, &nt, &maxi);                 // resolves the environment default
    const int prev = g_gn_fused_max_hw;
    if (fused_max_hw >= 0) g_gn_fused_max_hw = fused_max_hw;
    return prev;
}

size_t mve_groupnorm_workspace_bytes(int B, int HW, int C, int G) {
    if (B <= 0 || HW <= 0 || C <= 0) return 64;
    const int ns = gn_nsplit(B, HW, C);
    return sizeof(float) * ((size_t)B * ns * C * 2 + (size_t)B * G * 2 + 16);
}

int mve_groupnorm_silu(int dtype, const void* x1, int C1, const void* x2, int C2, int B, int HW, int G, float eps,
                       const float* gamma, const float* beta, int silu, void* out, void* workspace, void* stream) {
    return mve_groupnorm_silu_pair(dtype, x1, C1, x2, C2, B, HW, G, eps, gamma, beta, silu, out, workspace, nullptr, nullptr, stream);
}

int mve_groupnorm_silu_pair(int dtype, const void* x1, int C1, const void* x2, int C2, int B, int HW, int G, float eps,
                            const float* gamma, const float* beta, int silu, void* out, void* workspace, const void* x1_lo, const void* x2_lo,
                            void* stream) {
    if (B == 0 || HW == 0) return MVE_OK;
    const int C = C1 + C2;
    MVE_CHECK(x1 && out && gamma && beta && workspace && (C2 == 0 || x2), MVE_ERR_ARG, "groupnorm: null pointer");
    MVE_CHECK(C1 % 8 == 0 && C2 % 8 == 0 && G > 0 && C % G == 0, MVE_ERR_ARG,
              "groupnorm: C1=%d C2=%d must be multiples of 8 and C divisible by G=%d", C1, C2, G);
    GNSrc s;
    s.x1 = x1; s.x2 = x2; s.C1 = C1; s.C2 = C2; s.HW = HW; s.B = B;
    s.x1_lo = x1_lo; s.x2_lo = C2 ? x2_lo : nullptr;
    if (dtype == MVE_F16) return gn_run<F16Tag>(s, G, eps, gamma, beta, silu, out, (float*)workspace, (hipStream_t)stream);
    if (dtype == MVE_BF16) return gn_run<BF16Tag>(s, G, eps, gamma, beta, silu, out, (float*)workspace, (hipStream_t)stream);
    mve_set_error("groupnorm: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

int mve_layernorm(int dtype, const void* x, int ldx, void* y, int ldy, int M, int C, const float* gamma, const float* beta,
                  float eps, void* stream) {
    return mve_layernorm_pair(dtype, x, ldx, y, ldy, M, C, gamma, beta, eps, nullptr, stream);
}

int mve_layernorm_pair(int dtype, const void* x, int ldx, void* y, int ldy, int M, int C, const float* gamma, const float* beta,
                       float eps, const void* x_lo, void* stream) {
    if (M == 0) return MVE_OK;
    MVE_CHECK(x && y && gamma && beta, MVE_ERR_ARG, "layernorm: null pointer");
    MVE_CHECK(C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && ldy % 8 == 0, MVE_ERR_ARG,
              "layernorm: C=%d must be a multiple of 8 and <= 2048", C);
    if (dtype == MVE_F16) return ln_run<F16Tag>(x, ldx, y, ldy, M, C, gamma, beta, eps, (hipStream_t)stream, x_lo);
    if (dtype == MVE_BF16) return ln_run<BF16Tag>(x, ldx, y, ldy, M, C, gamma, beta, eps, (hipStream_t)stream, x_lo);
    mve_set_error("layernorm: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

}  // extern "C"
