// Tri-plane radiance decoders (gfx950; FMA-bound per-point MLP, gather-bound plane fetches): TriPlaneDecoder.point_decode
// (lib/models/decoders/triplane_decoder.py:135-199) and TriPlaneiNGPDecoder.point_decode (lib/models/decoders/triplane_ingp_decoder.py:
// 142-212), forward.  One wave = 64 points.  The first layer is accumulated feature by feature as the plane fetches (and hash-grid levels)
// produce them -- the 96-wide feature vector never exists; weights are wave-uniform and transposed ([in][out]), so one input's fan-out is
// one contiguous scalar load; the activated hidden vector goes through LDS ([k][lane]: conflict-free) so that the second layer can loop over
// its inputs at run time with all its outputs in registers.  The hash-grid encoding is the one of nerf.hip (hashgrid.h).
#include "raymarch_core.h"

#include "hashgrid.h"
#include "sh_core.h"

namespace {

constexpr int WAVE = 64;

struct ShK16 { float k[MVE_SH_MAX_DEGREE * MVE_SH_MAX_DEGREE]; };

struct TriParams {
    const float *xyz, *dirs, *code;
    int N, C, h, w;
    int axes[6];
    int flip_z;
    const float *base_wT, *base_b, *ingp_wT, *ingp_b, *dens_w, *dens_b, *col1_wT, *col1_b, *col2_w, *col2_b;
    int activation, sigma_activation;
    float sat;
    float *sigmas, *rgbs;
    DecoderParams hg;         // table / level meta / bound of the hash grid (only .table, .bound, .g are used)
};

__device__ __forceinline__ float act_fn(int a, float x) {
    if (a == 0) return fmaxf(x, 0.f);
    if (a == 1) return x / (1.0f + __expf(-x));
    if (a == 2) return x > 20.f ? x : log1pf(__expf(x));          // torch.nn.Softplus (beta 1, threshold 20)
    return __expf(x);                                              // trunc_exp forward
}

template <int H, int H2, int NL>
__global__ __launch_bounds__(WAVE) void k_triplane_decode(const TriParams p, const ShK16 kk) {
    __shared__ float hid[(H + 16) * WAVE];            // activated hidden vector + SH encoding, [k][lane]
    const int lane = threadIdx.x;
    const int i = blockIdx.x * WAVE + lane;
    const bool live = i < p.N;
    const int ii = live ? i : p.N - 1;
    const float x[3] = {p.xyz[3 * ii], p.xyz[3 * ii + 1], p.flip_z ? -p.xyz[3 * ii + 2] : p.xyz[3 * ii + 2]};
    float acc[H];
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = p.base_b[j];
    // ---- plane features, accumulated into the first layer as they come -----------------------------------------------------
    for (int pl = 0; pl < 3; ++pl) {
        const float u = x[p.axes[2 * pl]], v = x[p.axes[2 * pl + 1]];
        // F.grid_sample(align_corners=False, padding_mode='border'): unnormalise, clip to [0, size - 1], bilinear
        float fx = ((u + 1.0f) * (float)p.w - 1.0f) * 0.5f, fy = ((v + 1.0f) * (float)p.h - 1.0f) * 0.5f;
        fx = fminf(fmaxf(fx, 0.f), (float)(p.w - 1));
        fy = fminf(fmaxf(fy, 0.f), (float)(p.h - 1));
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = x0 + 1 < p.w ? x0 + 1 : p.w - 1, y1 = y0 + 1 < p.h ? y0 + 1 : p.h - 1;     // weight 0 where clipped
        const float tx = fx - x0f, ty = fy - y0f;
        const float w00 = (1.f - tx) * (1.f - ty), w10 = tx * (1.f - ty), w01 = (1.f - tx) * ty, w11 = tx * ty;
        const float* pc = p.code + (size_t)pl * p.h * p.w * p.C;
        const float* c00 = pc + ((size_t)y0 * p.w + x0) * p.C;
        const float* c10 = pc + ((size_t)y0 * p.w + x1) * p.C;
        const float* c01 = pc + ((size_t)y1 * p.w + x0) * p.C;
        const float* c11 = pc + ((size_t)y1 * p.w + x1) * p.C;
        for (int c = 0; c < p.C; ++c) {
            // torch's bilinear kernel sums nw, ne, sw, se in this order
            const float f = c00[c] * w00 + c10[c] * w10 + c01[c] * w01 + c11[c] * w11;
            const float* wr = p.base_wT + (size_t)(c * 3 + pl) * H;
#pragma unroll
            for (int j = 0; j < H; ++j) acc[j] = fmaf(wr[j], f, acc[j]);
        }
    }
    if constexpr (NL > 0) {
        float enc[2 * NL];
        hash_encode<NL>(p.hg, x[0], x[1], p.flip_z ? -x[2] : x[2], enc);          // the hash grid sees the un-flipped point
#pragma unroll
        for (int e = 0; e < 2 * NL; ++e) {
            const float* wr = p.ingp_wT + (size_t)e * H;
#pragma unroll
            for (int j = 0; j < H; ++j) acc[j] = fmaf(wr[j], enc[e], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < H; ++j) acc[j] += p.ingp_b[j];
    }
    // ---- density head; activated hidden vector to LDS ------------------------------------------------------------------------
    float sg = p.dens_b[0];
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const float a = act_fn(p.activation, acc[j]);
        hid[j * WAVE + lane] = a;
        sg = fmaf(p.dens_w[j], a, sg);
    }
    if (live) p.sigmas[i] = act_fn(p.sigma_activation, sg);
    if (p.dirs == nullptr) return;
    {
        float sh[16];
        she_eval(kk.k, p.dirs[3 * ii], p.dirs[3 * ii + 1], p.dirs[3 * ii + 2], 4, sh, nullptr);
#pragma unroll
        for (int k = 0; k < 16; ++k) hid[(H + k) * WAVE + lane] = sh[k];
    }
    // ---- colour net: Linear(H + 16 -> H2), act, Linear(H2 -> 3), sigmoid, saturation -------------------------------------------
    float a2[H2];
#pragma unroll
    for (int j = 0; j < H2; ++j) a2[j] = p.col1_b[j];
    for (int k = 0; k < H + 16; ++k) {
        const float a = hid[k * WAVE + lane];
        const float* wr = p.col1_wT + (size_t)k * H2;
#pragma unroll
        for (int j = 0; j < H2; ++j) a2[j] = fmaf(wr[j], a, a2[j]);
    }
    float o[3] = {p.col2_b[0], p.col2_b[1], p.col2_b[2]};
#pragma unroll
    for (int j = 0; j < H2; ++j) {
        const float a = act_fn(p.activation, a2[j]);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = fmaf(p.col2_w[c * H2 + j], a, o[c]);
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p.rgbs[3 * i + c] = (1.0f / (1.0f + __expf(-o[c]))) * (1.0f + 2.0f * p.sat) - p.sat;
    }
}

// d act(x) / dx for the hidden activations and the density head (trunc_exp: lib/ops/activation.py:17-20, g * clamp(exp(x), 1e-6, 1e6))
__device__ __forceinline__ float act_grad(int a, float x) {
    if (a == 0) return x > 0.f ? 1.f : 0.f;
    if (a == 1) { const float s = 1.0f / (1.0f + __expf(-x)); return s * (1.0f + x * (1.0f - s)); }
    if (a == 2) return x > 20.f ? 1.f : 1.0f / (1.0f + __expf(-x));
    return fminf(fmaxf(__expf(x), 1e-6f), 1e6f);
}

// Backward of k_triplane_decode (what autograd supplies when the reference optimises a TriPlane(iNGP)Decoder scene inside nerf_optim,
// lib/pipelines/mvedit_3d_pipeline.py:507-633): gradients of sum(g_sigma * sigma) + sum(g_rgb * rgb) w.r.t. the code planes, the hash table
// and every Linear.  One wave = 64 points, forward recomputed (nothing is saved).  The per-point factors of the weight gradients go to a
// workspace and k_tri_xty reduces them in a fixed order (deterministic); the plane and table gradients are scattered with float atomics, as
// F.grid_sample's and tiny-cuda-nn's backward do.
struct TriBwd {
    const float *g_sigma, *g_rgb;
    float *g_code, *g_table;
    float *ws_feat, *ws_enc, *ws_dbase, *ws_hid, *ws_da2, *ws_act2, *ws_do;      // [N][3C], [N][2NL], [N][H], [N][H+16], [N][H2], [N][H2], [N][4]
};

template <int H, int H2, int NL>
__global__ __launch_bounds__(WAVE) void k_triplane_backward(const TriParams p, const TriBwd b, const ShK16 kk) {
    __shared__ float hid[(H + 16) * WAVE];
    __shared__ float da2s[H2 * WAVE];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * WAVE + lane;
    const bool live = i < p.N;
    const int ii = live ? i : p.N - 1;
    const float x[3] = {p.xyz[3 * ii], p.xyz[3 * ii + 1], p.flip_z ? -p.xyz[3 * ii + 2] : p.xyz[3 * ii + 2]};
    const int F3 = 3 * p.C;
    float acc[H];
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = p.base_b[j];
    // plane taps of this point: kept for the scatter at the end
    int tx0[3], tx1[3], ty0[3], ty1[3];
    float tw[3][4];
    for (int pl = 0; pl < 3; ++pl) {
        const float u = x[p.axes[2 * pl]], v = x[p.axes[2 * pl + 1]];
        float fx = ((u + 1.0f) * (float)p.w - 1.0f) * 0.5f, fy = ((v + 1.0f) * (float)p.h - 1.0f) * 0.5f;
        fx = fminf(fmaxf(fx, 0.f), (float)(p.w - 1));
        fy = fminf(fmaxf(fy, 0.f), (float)(p.h - 1));
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = x0 + 1 < p.w ? x0 + 1 : p.w - 1, y1 = y0 + 1 < p.h ? y0 + 1 : p.h - 1;
        const float tx = fx - x0f, ty = fy - y0f;
        tx0[pl] = x0; tx1[pl] = x1; ty0[pl] = y0; ty1[pl] = y1;
        tw[pl][0] = (1.f - tx) * (1.f - ty); tw[pl][1] = tx * (1.f - ty); tw[pl][2] = (1.f - tx) * ty; tw[pl][3] = tx * ty;
        const float* pc = p.code + (size_t)pl * p.h * p.w * p.C;
        const float* c00 = pc + ((size_t)y0 * p.w + x0) * p.C;
        const float* c10 = pc + ((size_t)y0 * p.w + x1) * p.C;
        const float* c01 = pc + ((size_t)y1 * p.w + x0) * p.C;
        const float* c11 = pc + ((size_t)y1 * p.w + x1) * p.C;
        for (int c = 0; c < p.C; ++c) {
            const float f = c00[c] * tw[pl][0] + c10[c] * tw[pl][1] + c01[c] * tw[pl][2] + c11[c] * tw[pl][3];
            if (live) b.ws_feat[(size_t)i * F3 + c * 3 + pl] = f;
            const float* wr = p.base_wT + (size_t)(c * 3 + pl) * H;
#pragma unroll
            for (int j = 0; j < H; ++j) acc[j] = fmaf(wr[j], f, acc[j]);
        }
    }
    if constexpr (NL > 0) {
        float enc[2 * NL];
        hash_encode<NL>(p.hg, x[0], x[1], p.flip_z ? -x[2] : x[2], enc);
#pragma unroll
        for (int e = 0; e < 2 * NL; ++e) {
            if (live) b.ws_enc[(size_t)i * (2 * NL) + e] = enc[e];
            const float* wr = p.ingp_wT + (size_t)e * H;
#pragma unroll
            for (int j = 0; j < H; ++j) acc[j] = fmaf(wr[j], enc[e], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < H; ++j) acc[j] += p.ingp_b[j];
    }
    float sg = p.dens_b[0];
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const float a = act_fn(p.activation, acc[j]);
        hid[j * WAVE + lane] = a;
        sg = fmaf(p.dens_w[j], a, sg);
    }
    const float d_sg = live ? b.g_sigma[i] * act_grad(p.sigma_activation, sg) : 0.f;
    float d_o[3] = {0.f, 0.f, 0.f};
    const bool colour = p.dirs != nullptr && b.g_rgb != nullptr;
    if (colour) {
        float sh[16];
        she_eval(kk.k, p.dirs[3 * ii], p.dirs[3 * ii + 1], p.dirs[3 * ii + 2], 4, sh, nullptr);
#pragma unroll
        for (int k = 0; k < 16; ++k) hid[(H + k) * WAVE + lane] = sh[k];
        float a2[H2];
#pragma unroll
        for (int j = 0; j < H2; ++j) a2[j] = p.col1_b[j];
        for (int k = 0; k < H + 16; ++k) {
            const float a = hid[k * WAVE + lane];
            const float* wr = p.col1_wT + (size_t)k * H2;
#pragma unroll
            for (int j = 0; j < H2; ++j) a2[j] = fmaf(wr[j], a, a2[j]);
        }
        float o[3] = {p.col2_b[0], p.col2_b[1], p.col2_b[2]};
#pragma unroll
        for (int j = 0; j < H2; ++j) {
            const float a = act_fn(p.activation, a2[j]);
            if (live) b.ws_act2[(size_t)i * H2 + j] = a;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = fmaf(p.col2_w[c * H2 + j], a, o[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = 1.0f / (1.0f + __expf(-o[c]));
            d_o[c] = live ? b.g_rgb[3 * (size_t)i + c] * (1.0f + 2.0f * p.sat) * s * (1.0f - s) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < H2; ++j) {
            float g = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) g = fmaf(p.col2_w[c * H2 + j], d_o[c], g);
            g *= act_grad(p.activation, a2[j]);
            da2s[j * WAVE + lane] = g;
            if (live) b.ws_da2[(size_t)i * H2 + j] = g;
        }
    }
    if (live) {
        reinterpret_cast<f32x4*>(b.ws_do)[i] = f32x4{d_o[0], d_o[1], d_o[2], d_sg};
        for (int k = 0; k < H + 16; ++k) b.ws_hid[(size_t)i * (H + 16) + k] = (k < H || colour) ? hid[k * WAVE + lane] : 0.f;
    }
    // d base_x[k] = (dens_w[k] d_sg + sum_j col1_wT[k][j] d_a2[j]) * act'(base_x[k]); acc becomes d base_x
#pragma unroll
    for (int k = 0; k < H; ++k) {
        float g = p.dens_w[k] * d_sg;
        if (colour) {
            const float* wr = p.col1_wT + (size_t)k * H2;
            for (int j = 0; j < H2; ++j) g = fmaf(wr[j], da2s[j * WAVE + lane], g);
        }
        g *= act_grad(p.activation, acc[k]);
        acc[k] = g;
        if (live) b.ws_dbase[(size_t)i * H + k] = g;
    }
    if (!live) return;
    // ---- code planes: d feature (c, plane) = base_wT[c * 3 + plane][:] . d base_x, scattered with the bilinear weights ----------------------
    if (b.g_code) {
        for (int pl = 0; pl < 3; ++pl) {
            float* gp = b.g_code + (size_t)pl * p.h * p.w * p.C;
            for (int c = 0; c < p.C; ++c) {
                const float* wr = p.base_wT + (size_t)(c * 3 + pl) * H;
                float g = 0.f;
#pragma unroll
                for (int j = 0; j < H; ++j) g = fmaf(wr[j], acc[j], g);
                atomicAdd(gp + ((size_t)ty0[pl] * p.w + tx0[pl]) * p.C + c, g * tw[pl][0]);
                atomicAdd(gp + ((size_t)ty0[pl] * p.w + tx1[pl]) * p.C + c, g * tw[pl][1]);
                atomicAdd(gp + ((size_t)ty1[pl] * p.w + tx0[pl]) * p.C + c, g * tw[pl][2]);
                atomicAdd(gp + ((size_t)ty1[pl] * p.w + tx1[pl]) * p.C + c, g * tw[pl][3]);
            }
        }
    }
    if constexpr (NL > 0) {
        if (b.g_table) {
            float denc[2 * NL];
#pragma unroll
            for (int e = 0; e < 2 * NL; ++e) {
                const float* wr = p.ingp_wT + (size_t)e * H;
                float g = 0.f;
#pragma unroll
                for (int j = 0; j < H; ++j) g = fmaf(wr[j], acc[j], g);
                denc[e] = g;
            }
            hash_scatter<NL>(p.hg, x[0], x[1], p.flip_z ? -x[2] : x[2], denc, b.g_table);
        }
    }
}

// out[pi][qi] = sum_m X[m][pi] Y[m][qi] and colsum[pi] = sum_m X[m][pi], for column windows of two row-major workspaces (leading dimensions
// ldx, ldy): per-block partials over 1024-row chunks, folded in block order by k_tri_xty_reduce -- deterministic.  P <= 64, Q <= 64, P * Q <= 2048.
constexpr int TXR = 1024, TXS = 32;
__global__ __launch_bounds__(256) void k_tri_xty(const float* __restrict__ X, int ldx, int P, const float* __restrict__ Y, int ldy, int Q, int M,
                                                 float* __restrict__ partial /*[nblk][P*Q + P]*/) {
    __shared__ float xs[TXS][64], ys[TXS][64];
    const int r0 = blockIdx.x * TXR, r1 = min(M, r0 + TXR);
    const int nout = P * Q;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float cs = 0.f;
    for (int r = r0; r < r1; r += TXS) {
        for (int t = threadIdx.x; t < TXS * P; t += 256) { const int rr = t / P, c = t - rr * P; xs[rr][c] = (r + rr < r1) ? X[(size_t)(r + rr) * ldx + c] : 0.f; }
        for (int t = threadIdx.x; t < TXS * Q; t += 256) { const int rr = t / Q, c = t - rr * Q; ys[rr][c] = (r + rr < r1) ? Y[(size_t)(r + rr) * ldy + c] : 0.f; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int oi = threadIdx.x + 256 * k;
            if (oi < nout) {
                const int pi = oi / Q, qi = oi - pi * Q;
                float a = acc[k];
                for (int rr = 0; rr < TXS; ++rr) a = fmaf(xs[rr][pi], ys[rr][qi], a);
                acc[k] = a;
            }
        }
        if ((int)threadIdx.x < P) for (int rr = 0; rr < TXS; ++rr) cs += xs[rr][threadIdx.x];
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * (nout + P);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int oi = threadIdx.x + 256 * k; if (oi < nout) out[oi] = acc[k]; }
    if ((int)threadIdx.x < P) out[nout + threadIdx.x] = cs;
}
// out_mat[pi * ldo + qi] (a [P][Q] window of a row-major matrix), out_col[pi]
__global__ __launch_bounds__(256) void k_tri_xty_reduce(const float* __restrict__ partial, int nblk, int P, int Q, float* __restrict__ out_mat, int ldo,
                                                        float* __restrict__ out_col) {
    const int t = blockIdx.x * 256 + threadIdx.x, n = P * Q + P;
    if (t >= n) return;
    float a = 0.f;
    for (int bb = 0; bb < nblk; ++bb) a += partial[(size_t)bb * n + t];
    if (t < P * Q) { const int pi = t / Q, qi = t - pi * Q; out_mat[(size_t)pi * ldo + qi] = a; }
    else if (out_col) out_col[t - P * Q] = a;
}

// G [P][Q] (leading dimension ldo) = X^T Y over M rows, in column windows that fit the reduction kernel; colsum(X) once
int tri_xty(const float* X, int ldx, int P, const float* Y, int ldy, int Q, int M, float* G, int ldo, float* colsum, float* partial, hipStream_t s) {
    const int nblk = (M + TXR - 1) / TXR;
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int pc = P - p0 < 64 ? P - p0 : 64;
        const int qstep = 2048 / pc < 64 ? 2048 / pc : 64;
        for (int q0 = 0; q0 < Q; q0 += qstep) {
            const int qc = Q - q0 < qstep ? Q - q0 : qstep;
            k_tri_xty<<<nblk, 256, 0, s>>>(X + p0, ldx, pc, Y + q0, ldy, qc, M, partial);
            MVE_LAUNCH_CHECK();
            k_tri_xty_reduce<<<mve_cdiv(pc * qc + pc, 256), 256, 0, s>>>(partial, nblk, pc, qc, G + (size_t)p0 * ldo + q0, ldo,
                                                                      (q0 == 0 && colsum) ? colsum + p0 : nullptr);
            MVE_LAUNCH_CHECK();
        }
    }
    return MVE_OK;
}

template <int H, int H2>
int launch_bwd_nl(const TriParams& p, const TriBwd& b, const ShK16& kk, int n_levels, hipStream_t s) {
    const unsigned grid = mve_cdiv((unsigned)p.N, WAVE);
    switch (n_levels) {
        case 0: k_triplane_backward<H, H2, 0><<<grid, WAVE, 0, s>>>(p, b, kk); break;
        case 12: k_triplane_backward<H, H2, 12><<<grid, WAVE, 0, s>>>(p, b, kk); break;
        case 14: k_triplane_backward<H, H2, 14><<<grid, WAVE, 0, s>>>(p, b, kk); break;
        case 16: k_triplane_backward<H, H2, 16><<<grid, WAVE, 0, s>>>(p, b, kk); break;
        default: mve_set_error("triplane_backward: n_levels must be 0 (no hash grid), 12, 14 or 16 (got %d)", n_levels); return MVE_ERR_ARG;
    }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

template <int H, int H2>
int launch_nl(const TriParams& p, const ShK16& kk, int n_levels, hipStream_t s) {
    const unsigned grid = mve_cdiv((unsigned)p.N, WAVE);
    switch (n_levels) {
        case 0: k_triplane_decode<H, H2, 0><<<grid, WAVE, 0, s>>>(p, kk); break;
        case 12: k_triplane_decode<H, H2, 12><<<grid, WAVE, 0, s>>>(p, kk); break;
        case 14: k_triplane_decode<H, H2, 14><<<grid, WAVE, 0, s>>>(p, kk); break;
        case 16: k_triplane_decode<H, H2, 16><<<grid, WAVE, 0, s>>>(p, kk); break;
        default: mve_set_error("triplane_decode: n_levels must be 0 (no hash grid), 12, 14 or 16 (got %d)", n_levels); return MVE_ERR_ARG;
    }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // namespace

namespace {

// validation shared by the forward and the backward; fills p, returns the number of hash-grid levels through *nl_out
int fill_tri(const char* who, const MveTriplaneDesc* d, TriParams& p, int* nl_out) {
    MVE_CHECK(d->d_xyz && d->d_code && d->d_base_wT && d->d_base_b && d->d_dens_w && d->d_dens_b, MVE_ERR_ARG, "%s: null pointer", who);
    MVE_CHECK(d->N > 0 && d->C > 0 && d->h > 0 && d->w > 0, MVE_ERR_ARG, "%s: bad sizes N=%d C=%d h=%d w=%d", who, d->N, d->C, d->h, d->w);
    MVE_CHECK((d->hidden == 64 || d->hidden == 128) && (d->hidden2 == 64 || d->hidden2 == 128), MVE_ERR_ARG,
              "%s: hidden widths must be 64 or 128 (got %d, %d)", who, d->hidden, d->hidden2);
    MVE_CHECK(d->activation >= 0 && d->activation <= 2 && d->sigma_activation >= 0 && d->sigma_activation <= 3, MVE_ERR_ARG, "%s: bad activation code", who);
    for (int k = 0; k < 6; ++k) MVE_CHECK(d->axes[k] >= 0 && d->axes[k] < 3, MVE_ERR_ARG, "%s: bad plane axis %d", who, d->axes[k]);
    if (d->d_dirs) MVE_CHECK(d->d_col1_wT && d->d_col1_b && d->d_col2_w && d->d_col2_b, MVE_ERR_ARG, "%s: null colour-net pointer", who);
    memset(&p, 0, sizeof(p));
    p.xyz = d->d_xyz; p.dirs = d->d_dirs; p.code = d->d_code; p.N = d->N; p.C = d->C; p.h = d->h; p.w = d->w;
    for (int k = 0; k < 6; ++k) p.axes[k] = d->axes[k];
    p.flip_z = d->flip_z;
    p.base_wT = d->d_base_wT; p.base_b = d->d_base_b; p.ingp_wT = d->d_ingp_wT; p.ingp_b = d->d_ingp_b;
    p.dens_w = d->d_dens_w; p.dens_b = d->d_dens_b; p.col1_wT = d->d_col1_wT; p.col1_b = d->d_col1_b; p.col2_w = d->d_col2_w; p.col2_b = d->d_col2_b;
    p.activation = d->activation; p.sigma_activation = d->sigma_activation; p.sat = d->sigmoid_saturation;
    p.sigmas = d->d_sigmas; p.rgbs = d->d_rgbs;
    int nl = 0;
    if (d->d_ingp_wT) {
        nl = d->n_levels;
        MVE_CHECK(d->d_ingp_b && d->d_table && d->level_scale && d->level_res && d->level_offset && d->level_size && nl > 0 && nl <= MAX_LEVELS, MVE_ERR_ARG,
                  "%s: incomplete hash-grid description", who);
        p.hg.table = d->d_table; p.hg.bound = d->bound; p.hg.g.n_levels = nl;
        for (int l = 0; l < nl; ++l) { p.hg.g.scale[l] = d->level_scale[l]; p.hg.g.res[l] = d->level_res[l]; p.hg.g.off[l] = d->level_offset[l]; p.hg.g.size[l] = d->level_size[l]; }
    }
    *nl_out = nl;
    return MVE_OK;
}

size_t tri_ws_floats(size_t N, int C, int H, int H2, int nl) {
    const size_t nblk = (N + TXR - 1) / TXR;
    return N * (size_t)(3 * C + 2 * nl + H + (H + 16) + H2 + H2 + 4) + nblk * (size_t)(2048 + 64) + 64;
}

}  // namespace

extern "C" int mve_triplane_decode(const MveTriplaneDesc* d, void* stream) {
    MVE_CHECK(d, MVE_ERR_ARG, "triplane_decode: null descriptor");
    if (d->N == 0) return MVE_OK;
    TriParams p;
    int nl = 0;
    if (int rc = fill_tri("triplane_decode", d, p, &nl)) return rc;
    MVE_CHECK(d->d_sigmas && (!d->d_dirs || d->d_rgbs), MVE_ERR_ARG, "triplane_decode: null output pointer");
    ShK16 kk;
    she_constants(kk.k);
    hipStream_t s = (hipStream_t)stream;
    if (d->hidden == 128 && d->hidden2 == 128) return launch_nl<128, 128>(p, kk, nl, s);
    if (d->hidden == 128 && d->hidden2 == 64) return launch_nl<128, 64>(p, kk, nl, s);
    if (d->hidden == 64 && d->hidden2 == 128) return launch_nl<64, 128>(p, kk, nl, s);
    return launch_nl<64, 64>(p, kk, nl, s);
}

extern "C" size_t mve_triplane_backward_workspace_bytes(int N, int C, int hidden, int hidden2, int n_levels) {
    return sizeof(float) * tri_ws_floats((size_t)(N > 0 ? N : 0), C, hidden, hidden2, n_levels);
}

extern "C" int mve_triplane_backward(const MveTriplaneDesc* d, const float* d_grad_sigmas, const float* d_grad_rgbs, const MveTriplaneGrads* g,
                                     void* d_workspace, size_t workspace_bytes, void* stream) {
    MVE_CHECK(d && g, MVE_ERR_ARG, "triplane_backward: null descriptor");
    MVE_CHECK(g->d_base_w && g->d_base_b && g->d_dens_w && g->d_dens_b, MVE_ERR_ARG, "triplane_backward: null gradient output");
    hipStream_t s = (hipStream_t)stream;
    const int H = d->hidden, H2 = d->hidden2, F3 = 3 * d->C;
    const bool colour = d->d_dirs != nullptr && d_grad_rgbs != nullptr;
    if (colour) MVE_CHECK(g->d_col1_w && g->d_col1_b && g->d_col2_w && g->d_col2_b, MVE_ERR_ARG, "triplane_backward: null colour-net gradient output");
    if (d->N == 0) return MVE_OK;
    TriParams p;
    int nl = 0;
    if (int rc = fill_tri("triplane_backward", d, p, &nl)) return rc;
    if (nl) MVE_CHECK(g->d_ingp_w && g->d_ingp_b, MVE_ERR_ARG, "triplane_backward: null hash-branch gradient output");
    MVE_CHECK(d_grad_sigmas && d_workspace, MVE_ERR_ARG, "triplane_backward: null pointer");
    MVE_CHECK(workspace_bytes >= mve_triplane_backward_workspace_bytes(d->N, d->C, H, H2, nl), MVE_ERR_NOMEM, "triplane_backward: workspace too small");
    const size_t N = (size_t)d->N;
    float* ws = (float*)d_workspace;
    TriBwd b;
    b.g_sigma = d_grad_sigmas; b.g_rgb = colour ? d_grad_rgbs : nullptr;
    b.g_code = g->d_code; b.g_table = nl ? g->d_table : nullptr;
    MVE_CHECK(((uintptr_t)d_workspace & 15) == 0, MVE_ERR_ARG, "triplane_backward: the workspace must be 16-byte aligned");
    b.ws_do = ws; ws += N * 4;            // first: it is written as f32x4 (16-byte stores) -- behind N * (3C + ...) floats it was only 8-byte aligned for odd N, C = 6
    b.ws_feat = ws; ws += N * F3;
    b.ws_enc = ws; ws += N * 2 * nl;
    b.ws_dbase = ws; ws += N * H;
    b.ws_hid = ws; ws += N * (H + 16);
    b.ws_da2 = ws; ws += N * H2;
    b.ws_act2 = ws; ws += N * H2;
    ws = (float*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
    float* partial = ws;
    if (!colour) p.dirs = nullptr;
    ShK16 kk;
    she_constants(kk.k);
    int rc;
    if (H == 128 && H2 == 128) rc = launch_bwd_nl<128, 128>(p, b, kk, nl, s);
    else if (H == 128 && H2 == 64) rc = launch_bwd_nl<128, 64>(p, b, kk, nl, s);
    else if (H == 64 && H2 == 128) rc = launch_bwd_nl<64, 128>(p, b, kk, nl, s);
    else rc = launch_bwd_nl<64, 64>(p, b, kk, nl, s);
    if (rc) return rc;
    const int M = d->N;
    // torch layouts: base_net.0.weight [H][3C] = d base^T feat, bias = colsum(d base); ingp likewise; density_net.0.weight [1][H] = d_sg^T act(base)
    if ((rc = tri_xty(b.ws_dbase, H, H, b.ws_feat, F3, F3, M, g->d_base_w, F3, g->d_base_b, partial, s))) return rc;
    if (nl) {
        if ((rc = tri_xty(b.ws_dbase, H, H, b.ws_enc, 2 * nl, 2 * nl, M, g->d_ingp_w, 2 * nl, g->d_ingp_b, partial, s))) return rc;
    }
    if ((rc = tri_xty(b.ws_do + 3, 4, 1, b.ws_hid, H + 16, H, M, g->d_dens_w, H, g->d_dens_b, partial, s))) return rc;
    if (colour) {
        if ((rc = tri_xty(b.ws_da2, H2, H2, b.ws_hid, H + 16, H + 16, M, g->d_col1_w, H + 16, g->d_col1_b, partial, s))) return rc;
        if ((rc = tri_xty(b.ws_do, 4, 3, b.ws_act2, H2, H2, M, g->d_col2_w, H2, g->d_col2_b, partial, s))) return rc;
    }
    return MVE_OK;
}
