// NeRF render path for gfx950: hash-grid + MLP radiance decode, the fused march->decode->composite renderer,
// camera ray generation and the depth -> normal / depth normalisation image ops.
//
// Reference behaviour (Lakonik/MVEdit):
//   iNGPDecoder.point_decode                lib/models/decoders/ingp_decoder.py:106-120  (tinycudann HashGrid + 2-layer MLP)
//   VolumeRenderer.forward (eval branch)    lib/models/decoders/base_volume_renderer.py:264-329
//   BaseNeRF.render                          lib/models/autoencoders/base_nerf.py:489-556
//   get_ray_directions/get_rays/depth_to_normal/normalize_depth   lib/core/utils/geometry_utils.py:18-55,119-168
//
// MI355X-first design of the eval renderer.  The reference runs a HOST loop of up to 1024 rounds of
// {march_rays -> point_decode -> composite_rays -> boolean-mask compaction}; every round is three launches, 56 B per sample
// written and re-read through HBM, and a device->host sync for the data-dependent compaction.  Here one launch does the whole
// ray: each lane walks its ray through the occupancy bitfield (the same DDA as the marching operators, bit for bit), decodes
// every occupied sample in registers (12-14 hash levels x 8 corner gathers, then the 24..28 -> 64 -> 4 MLP with the weights
// as scalar operands), composites it, and stops when the ray leaves the volume or saturates (T < T_thresh, reference
// semantics T = 1 - sum w).  Samples never touch HBM; the only traffic is 24 B/ray in, 20 B/ray out and the hash-table
// gathers (48 MiB table: resident in the 256 MiB Infinity Cache).  With noise = 0 (eval mode) the chunked reference loop and
// this single walk visit exactly the same samples in the same order, so the per-ray accumulation order is identical.
// A wave takes 64 consecutive pixels of one image row, so neighbouring lanes march similar lengths (bounded divergence).
#include "raymarch_core.h"

#include "hashgrid.h"

namespace {

constexpr int NB = 256;
// sigma = exp(h0 + blob(x)), rgb = sigmoid(h1..3) * (1 + 2 sat) - sat     (ingp_decoder.py:100-118)
template <int NL>
__device__ __forceinline__ void decode_point(const DecoderParams& p, float x, float y, float z, float& sigma, float (&rgb)[3]) {
    float enc[2 * NL];
    hash_encode<NL>(p, x, y, z, enc);
    float o[4] = {p.b2[0], p.b2[1], p.b2[2], p.b2[3]};
    const int H = p.hidden;
    for (int j = 0; j < H; ++j) {                      // weights are wave-uniform: scalar loads, SGPR operands
        float a = p.b1[j];
        const float* wr = p.w1 + j * (2 * NL);
#pragma unroll
        for (int i = 0; i < 2 * NL; ++i) a = fmaf(wr[i], enc[i], a);
        a = fmaxf(a, 0.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaf(p.w2[k * H + j], a, o[k]);
    }
    const float d2 = fmaxf(x * x + y * y + z * z, 0.2f);
    const float blob = p.blob_density * expf(-d2 * p.blob_inv_2r2);
    sigma = expf(o[0] + blob);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = (1.0f / (1.0f + expf(-o[1 + k]))) * p.sat_scale - p.sat_shift;
}

// ---------------------------------------------------------------------------------------------------------------------
// Occupancy-friendly decoder (hidden == 64).  The straight-line version above unrolls all levels: ~20 k instructions, > 256 VGPRs,
// ONE wave per SIMD -- and the kernel is bound by the latency of its own dependent index arithmetic.  Here the level loop is
// rolled and the first MLP layer is accumulated level by level (h_j += w1[j][2l] e0; h_j += w1[j][2l+1] e1 -- the same FMA
// sequence per hidden unit, so the result is bit-identical); w1 is staged once per block in LDS as [level][hidden][2] and read
// with broadcast ds_read_b128.  64 accumulators + one level's gathers fit in 128 VGPRs: four waves per SIMD.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int HID = 64;

template <int NL>
__device__ __forceinline__ void stage_w1(const DecoderParams& p, float* __restrict__ w1t) {
    for (int i = threadIdx.x; i < HID * 2 * NL; i += NB) {
        const int j = i / (2 * NL), c = i - j * (2 * NL);        // w1[j][c], c = 2*l + f
        w1t[((c >> 1) * HID + j) * 2 + (c & 1)] = p.w1[i];
    }
}

template <int NL>
__device__ __forceinline__ void decode_point2(const DecoderParams& p, const float* __restrict__ w1t, float x, float y, float z, float& sigma,
                                              float (&rgb)[3]) {
    float h[HID];
#pragma unroll
    for (int j = 0; j < HID; ++j) h[j] = p.b1[j];
    const float inv = 1.0f / (2.0f * p.bound);
    const float u[3] = {(x + p.bound) * inv, (y + p.bound) * inv, (z + p.bound) * inv};
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
        const float scale = p.g.scale[l];
        const uint32_t res = p.g.res[l], size = p.g.size[l];
        uint32_t cell[3];
        float w[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float pos = fmaf(scale, u[d], 0.5f);
            const float fl = floorf(pos);
            cell[d] = (uint32_t)(int)fl;
            const float fr = pos - fl;
            w[d] = fr * fr * (3.0f - 2.0f * fr);
        }
        const bool s1 = res <= size;
        const bool s2 = s1 && (uint64_t)res * res <= size;
        const uint64_t stride3 = (uint64_t)res * res * (s2 ? res : 1u);
        const bool hashed = s2 ? (size < stride3) : true;
        const bool pow2 = (size & (size - 1u)) == 0u;
        const float2p* tab = reinterpret_cast<const float2p*>(p.table) + p.g.off[l];
        float a0 = 0.f, a1 = 0.f;
        auto corners = [&](auto index_of) {
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                float wt = 1.0f;
                uint32_t c[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    if (corner & (1 << d)) { wt = wt * w[d]; c[d] = cell[d] + 1u; }
                    else { wt = wt * (1.0f - w[d]); c[d] = cell[d]; }
                }
                const float2p f = tab[index_of(c)];
                a0 = fmaf(wt, f.x, a0);
                a1 = fmaf(wt, f.y, a1);
            }
        };
        if (hashed && pow2) {            // wave-uniform: 2^k-row hash table
            const uint32_t mask = size - 1u;
            corners([&](const uint32_t (&c)[3]) { return ((c[0] * 1u) ^ (c[1] * 2654435761u) ^ (c[2] * 805459861u)) & mask; });
        } else if (!hashed) {            // dense level: an in-range point indexes below 2 * size
            corners([&](const uint32_t (&c)[3]) {
                uint32_t idx = c[0] + c[1] * res + c[2] * res * res;
                if (idx >= size) idx -= size;
                if (idx >= size) idx %= size;
                return idx;
            });
        } else {
            corners([&](const uint32_t (&c)[3]) { return ((c[0] * 1u) ^ (c[1] * 2654435761u) ^ (c[2] * 805459861u)) % size; });
        }
        const f32x4* wl = reinterpret_cast<const f32x4*>(w1t + l * (HID * 2));
#pragma unroll
        for (int j2 = 0; j2 < HID / 2; ++j2) {
            const f32x4 wv = wl[j2];                               // (w[2j][e0], w[2j][e1], w[2j+1][e0], w[2j+1][e1]), broadcast
            h[2 * j2] = fmaf(wv[0], a0, h[2 * j2]);
            h[2 * j2] = fmaf(wv[1], a1, h[2 * j2]);
            h[2 * j2 + 1] = fmaf(wv[2], a0, h[2 * j2 + 1]);
            h[2 * j2 + 1] = fmaf(wv[3], a1, h[2 * j2 + 1]);
        }
    }
    float o[4] = {p.b2[0], p.b2[1], p.b2[2], p.b2[3]};
#pragma unroll
    for (int j = 0; j < HID; ++j) {
        const float a = fmaxf(h[j], 0.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaf(p.w2[k * HID + j], a, o[k]);
    }
    const float d2 = fmaxf(x * x + y * y + z * z, 0.2f);
    const float blob = p.blob_density * expf(-d2 * p.blob_inv_2r2);
    sigma = expf(o[0] + blob);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = (1.0f / (1.0f + expf(-o[1 + k]))) * p.sat_scale - p.sat_shift;
}

// ---------------------------------------------------------------------------------------------------------------------
// Decoder backward: gradients of  sum(g_sigma * sigma) + sum(g_rgb * rgb)  w.r.t. the hash table and
// the MLP, for the points of a training batch.  One thread per point recomputes the forward (nothing is saved by the forward
// pass), back-propagates through the activations (sigma: the reference's _trunc_exp backward, lib/ops/activation.py:17-20) and
// the two linear layers, scatters d enc into the table gradient (atomicAdd, as tiny-cuda-nn does) and leaves the per-point
// factors of the weight gradients (dh, relu(h), enc, do) in a workspace; k_xty reduces them deterministically.
// ---------------------------------------------------------------------------------------------------------------------
template <int NL>
__global__ __launch_bounds__(NB, 2) void k_decode_backward(DecoderParams p, const float* __restrict__ xyz, uint32_t M,
                                                           const float* __restrict__ g_sigma, const float* __restrict__ g_rgb,
                                                           float* __restrict__ g_table, float* __restrict__ ws_dh /*[M][64]*/,
                                                           float* __restrict__ ws_hr /*[M][64]*/, float* __restrict__ ws_enc /*[M][2NL]*/,
                                                           float* __restrict__ ws_do /*[M][4]*/) {
    __shared__ __attribute__((aligned(16))) float w1t[HID * 2 * NL];
    stage_w1<NL>(p, w1t);
    __syncthreads();
    // (every lane stays to the end: the table-gradient scatter below merges neighbouring lanes' contributions with cross-lane moves.  A lane past
    // the last point redoes point M - 1 with zero incoming gradients, stores nothing and scatters nothing.)
    const uint32_t m_raw = blockIdx.x * NB + threadIdx.x;
    const bool valid = m_raw < M;
    const uint32_t m = valid ? m_raw : M - 1;
    const float3p q = reinterpret_cast<const float3p*>(xyz)[m];
    const float x = q.x, y = q.y, z = q.z;
    const float inv = 1.0f / (2.0f * p.bound);
    const float u[3] = {(x + p.bound) * inv, (y + p.bound) * inv, (z + p.bound) * inv};
    // one level: cell / interpolation weights / table row of each corner, through `visit(corner_weight, row)`
    uint32_t cell_key = 0;                    // cell of the last level() call, 10 bits per axis (exact for resolutions below 1023)
    auto level = [&](int l, auto&& visit) {
        const float scale = p.g.scale[l];
        const uint32_t res = p.g.res[l], size = p.g.size[l];
        uint32_t cell[3];
        float w[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float pos = fmaf(scale, u[d], 0.5f);
            const float fl = floorf(pos);
            cell[d] = (uint32_t)(int)fl;
            const float fr = pos - fl;
            w[d] = fr * fr * (3.0f - 2.0f * fr);
        }
        cell_key = (cell[0] & 1023u) | ((cell[1] & 1023u) << 10) | ((cell[2] & 1023u) << 20);
        const bool s1 = res <= size;
        const bool s2 = s1 && (uint64_t)res * res <= size;
        const uint64_t stride3 = (uint64_t)res * res * (s2 ? res : 1u);
        const bool hashed = s2 ? (size < stride3) : true;
        const uint32_t off = p.g.off[l];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            float wt = 1.0f;
            uint32_t c[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (corner & (1 << d)) { wt = wt * w[d]; c[d] = cell[d] + 1u; }
                else { wt = wt * (1.0f - w[d]); c[d] = cell[d]; }
            }
            uint32_t idx = hashed ? ((c[0] * 1u) ^ (c[1] * 2654435761u) ^ (c[2] * 805459861u)) : (c[0] + c[1] * res + c[2] * res * res);
            idx %= size;
            visit(wt, off + idx);
        }
    };
    // ---- forward: hidden pre-activations -------------------------------------------------------------------------------
    float h[HID];
#pragma unroll
    for (int j = 0; j < HID; ++j) h[j] = p.b1[j];
    const float2p* tab = reinterpret_cast<const float2p*>(p.table);
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
        float a0 = 0.f, a1 = 0.f;
        level(l, [&](float wt, uint32_t row) { const float2p f = tab[row]; a0 = fmaf(wt, f.x, a0); a1 = fmaf(wt, f.y, a1); });
        if (valid) {
            ws_enc[(size_t)m * (2 * NL) + 2 * l] = a0;
            ws_enc[(size_t)m * (2 * NL) + 2 * l + 1] = a1;
        }
        const f32x4* wl = reinterpret_cast<const f32x4*>(w1t + l * (HID * 2));
#pragma unroll
        for (int j2 = 0; j2 < HID / 2; ++j2) {
            const f32x4 wv = wl[j2];
            h[2 * j2] = fmaf(wv[0], a0, h[2 * j2]);
            h[2 * j2] = fmaf(wv[1], a1, h[2 * j2]);
            h[2 * j2 + 1] = fmaf(wv[2], a0, h[2 * j2 + 1]);
            h[2 * j2 + 1] = fmaf(wv[3], a1, h[2 * j2 + 1]);
        }
    }
    float o[4] = {p.b2[0], p.b2[1], p.b2[2], p.b2[3]};
#pragma unroll
    for (int j = 0; j < HID; ++j) {
        const float a = fmaxf(h[j], 0.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaf(p.w2[k * HID + j], a, o[k]);
    }
    // ---- output activations backward -----------------------------------------------------------------------------------------
    const float d2 = fmaxf(x * x + y * y + z * z, 0.2f);
    const float sigma = expf(o[0] + p.blob_density * expf(-d2 * p.blob_inv_2r2));
    float dout[4];
    dout[0] = valid ? g_sigma[m] * fminf(fmaxf(sigma, 1e-6f), 1e6f) : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sg = 1.0f / (1.0f + expf(-o[1 + k]));
        dout[1 + k] = (valid && g_rgb ? g_rgb[3ull * m + k] : 0.0f) * p.sat_scale * sg * (1.0f - sg);
    }
    if (valid) reinterpret_cast<f32x4*>(ws_do)[m] = f32x4{dout[0], dout[1], dout[2], dout[3]};
    // ---- hidden layer backward: store relu(h) (for dW2) and dh (for dW1), keep dh in registers -------------------------------------
#pragma unroll
    for (int j4 = 0; j4 < HID / 4; ++j4) {
        f32x4 hr, dh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * j4 + e;
            const bool on = h[j] > 0.0f;
            hr[e] = on ? h[j] : 0.0f;
            float g = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) g = fmaf(p.w2[k * HID + j], dout[k], g);
            dh[e] = on ? g : 0.0f;
            h[j] = dh[e];
        }
        if (valid) {
            reinterpret_cast<f32x4*>(ws_hr + (size_t)m * HID)[j4] = hr;
            reinterpret_cast<f32x4*>(ws_dh + (size_t)m * HID)[j4] = dh;
        }
    }
    // ---- encoding backward: d enc_l = W1[:, 2l:2l+2]^T dh, scattered to the 8 corners of every level ------------------------------
    // (all levels' d enc first, parked in LDS: the 64 hidden-unit gradients are dead by the time the scatter loop needs its registers for the merge)
    __shared__ float es[2 * NL][NB];
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
        const f32x4* wl = reinterpret_cast<const f32x4*>(w1t + l * (HID * 2));
        float e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int j2 = 0; j2 < HID / 2; ++j2) {
            const f32x4 wv = wl[j2];
            e0 = fmaf(wv[0], h[2 * j2], e0); e1 = fmaf(wv[1], h[2 * j2], e1);
            e0 = fmaf(wv[2], h[2 * j2 + 1], e0); e1 = fmaf(wv[3], h[2 * j2 + 1], e1);
        }
        es[2 * l][threadIdx.x] = e0;
        es[2 * l + 1][threadIdx.x] = e1;
    }
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
        const float e0 = es[2 * l][threadIdx.x], e1 = es[2 * l + 1][threadIdx.x];      // (a thread reads back its own slots: no barrier)
        // Points are in ray order, so neighbouring lanes sit in the same cell of a coarse level (18 consecutive samples per 16^3 cell at the
        // renderer's step, 2 at 141^3) and would send the same 8 rows 2 x 8 float atomics each.  MI355X retires ~20.8 G float atomics / s whatever
        // the table size, the contention or the scope (tools/probe/atomic_probe.hip, profiles/r04_atomic_probe.log): the 46.6 M atomics of a
        // 242 k-point batch were 2.2 ms of this kernel's 3.5.  So lanes merge first: a butterfly over aligned blocks of 2, 4, 8, 16 lanes -- the
        // leader of a block takes its partner block's leader's sums when both hold the same cell (exact 30-bit cell key), and only lanes still
        // holding sums scatter.  (Summation order changes, as it does with every run of the atomics themselves.)
        // (four corners at a time: all sixteen sums next to the 64 hidden-unit gradients would spill)
        bool alive = valid;
        bool take[4] = {false, false, false, false};
        const bool merge = p.g.res[l] <= 160u;    // (wave-uniform: finer levels have ~1 point per cell, nothing to merge)
        const int lane = threadIdx.x & 63;
        {
            level(l, [&](float, uint32_t) {});    // the cell of this level -> cell_key
            const uint32_t key = cell_key;
            if (merge) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int d = 1 << s4;
                    const uint32_t pkey = (uint32_t)__shfl_xor((int)key, d, 64);
                    const int palive = __shfl_xor((int)alive, d, 64);
                    const bool can = ((lane & (d - 1)) == 0) && alive && palive && pkey == key;      // both lead their block of d lanes and hold the same cell
                    take[s4] = can && !(lane & d);
                    alive = alive && !(can && (lane & d));
                }
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float c0[4], c1[4];
            uint32_t rows[4];
            int k = 0;
            level(l, [&](float wt, uint32_t row) {
                if ((k >> 2) == half) { c0[k & 3] = wt * e0; c1[k & 3] = wt * e1; rows[k & 3] = row; }
                ++k;
            });
            if (merge) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int d = 1 << s4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float p0 = __shfl_xor(c0[j], d, 64), p1 = __shfl_xor(c1[j], d, 64);
                        c0[j] += take[s4] ? p0 : 0.0f;
                        c1[j] += take[s4] ? p1 : 0.0f;
                    }
                }
            }
            if (alive) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    atomicAdd(g_table + 2ull * rows[j], c0[j]);
                    atomicAdd(g_table + 2ull * rows[j] + 1, c1[j]);
                }
            }
        }
    }
}

// out[p][q] = sum_m X[m][p] * Y[m][q]  (and colsum: sum_m X[m][p]): per-block partials over XTY_ROWS-row chunks, reduced in a
// fixed order by k_xty_reduce -> deterministic.  P, Q <= 64, P * Q <= 2048.
constexpr int XTY_ROWS = 256, XTY_STEP = 32;       // (256-row chunks since round 4: 948 blocks for a 242 k-point batch instead of 237 on 256 CUs -- the kernel is LDS-read bound per block)
__global__ __launch_bounds__(256) void k_xty(const float* __restrict__ X, int P, const float* __restrict__ Y, int Q, uint32_t M,
                                             float* __restrict__ partial /*[nblk][P*Q + P]*/) {
    __shared__ float xs[XTY_STEP][64], ys[XTY_STEP][64];
    const uint32_t r0 = blockIdx.x * XTY_ROWS, r1 = min(M, r0 + XTY_ROWS);
    const int nout = P * Q;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float cs = 0.f;                                       // threads 0..P-1: column sums of X
    for (uint32_t r = r0; r < r1; r += XTY_STEP) {
        for (int i = threadIdx.x; i < XTY_STEP * P; i += 256) { const int rr = i / P, c = i - rr * P; xs[rr][c] = (r + rr < r1) ? X[(size_t)(r + rr) * P + c] : 0.f; }
        for (int i = threadIdx.x; i < XTY_STEP * Q; i += 256) { const int rr = i / Q, c = i - rr * Q; ys[rr][c] = (r + rr < r1) ? Y[(size_t)(r + rr) * Q + c] : 0.f; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int oi = threadIdx.x + 256 * k;
            if (oi < nout) {
                const int pi = oi / Q, qi = oi - pi * Q;
                float a = acc[k];
                for (int rr = 0; rr < XTY_STEP; ++rr) a = fmaf(xs[rr][pi], ys[rr][qi], a);
                acc[k] = a;
            }
        }
        if ((int)threadIdx.x < P) for (int rr = 0; rr < XTY_STEP; ++rr) cs += xs[rr][threadIdx.x];
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * (nout + P);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int oi = threadIdx.x + 256 * k; if (oi < nout) out[oi] = acc[k]; }
    if ((int)threadIdx.x < P) out[nout + threadIdx.x] = cs;
}
__global__ __launch_bounds__(256) void k_xty_reduce(const float* __restrict__ partial, int nblk, int n, int n_mat, float* __restrict__ out_mat,
                                                    float* __restrict__ out_col) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // eight independent chains (fixed order: deterministic): one chain waits out a load per partial
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] += partial[(size_t)(b + k) * n + i];
    }
    for (; b < nblk; ++b) a8[b & 7] += partial[(size_t)b * n + i];
    const float a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    if (i < n_mat) out_mat[i] = a;
    else if (out_col) out_col[i - n_mat] = a;
}

// fused Adam (torch.optim.Adam semantics, no weight decay / amsgrad): in place on param, m, v
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                                              size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float a = b1 * m1[i] + (1.0f - b1) * gi;
    const float b = b2 * m2[i] + (1.0f - b2) * gi * gi;
    m1[i] = a; m2[i] = b;
    w[i] -= lr * (a / bc1) / (sqrtf(b / bc2) + eps);
}

template <int NL>
__global__ __launch_bounds__(NB, 4) void k_point_decode2(DecoderParams p, const float* __restrict__ xyz, uint32_t M,
                                                         float* __restrict__ sigmas, float* __restrict__ rgbs) {
    __shared__ __attribute__((aligned(16))) float w1t[HID * 2 * NL];
    stage_w1<NL>(p, w1t);
    __syncthreads();
    const uint32_t i = blockIdx.x * NB + threadIdx.x;
    if (i >= M) return;
    const float3p q = reinterpret_cast<const float3p*>(xyz)[i];
    float s, c[3];
    decode_point2<NL>(p, w1t, q.x, q.y, q.z, s, c);
    sigmas[i] = s;
    if (rgbs) reinterpret_cast<float3p*>(rgbs)[i] = float3p{c[0], c[1], c[2]};
}

template <int NL>
__global__ __launch_bounds__(NB, 3) void k_render_rays2(DecoderParams dp, MarchParams mp, const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, const float* __restrict__ aabb,
                                                        uint32_t N, float min_near, float T_thresh,
                                                        float* __restrict__ weights_sum, float* __restrict__ depth,
                                                        float* __restrict__ image, int32_t* __restrict__ n_samples) {
    __shared__ __attribute__((aligned(16))) float w1t[HID * 2 * NL];
    stage_w1<NL>(dp, w1t);
    __syncthreads();
    const uint32_t n = blockIdx.x * NB + threadIdx.x;
    if (n >= N) return;
    const float3p o = reinterpret_cast<const float3p*>(rays_o)[n];
    const float3p d = reinterpret_cast<const float3p*>(rays_d)[n];
    float near, far;
    near_far_one(o, d, aabb, min_near, near, far);
    GridWalker w;
    w.init(mp, rays_o + 3ull * n, rays_d + 3ull * n);
    float ws = 0.f, dep = 0.f, r = 0.f, g = 0.f, b = 0.f;
    float t = near;
    t += w.step_len(t) * 0.0f;
    const uint32_t cnt = w.walk(t, far, mp.max_steps, [&](float cx, float cy, float cz, float tn, float dt) {
        float sigma, c[3];
        decode_point2<NL>(dp, w1t, cx, cy, cz, sigma, c);
        const float alpha = 1.0f - __expf(-sigma * dt);
        const float T = 1 - ws;
        const float wgt = alpha * T;
        ws += wgt;
        dep += wgt / tn;
        r += wgt * c[0];
        g += wgt * c[1];
        b += wgt * c[2];
        return !(T < T_thresh);
    });
    weights_sum[n] = ws;
    depth[n] = dep;
    reinterpret_cast<float3p*>(image)[n] = float3p{r, g, b};
    if (n_samples) n_samples[n] = (int32_t)cnt;
}

template <int NL>
__global__ __launch_bounds__(NB) void k_point_decode(DecoderParams p, const float* __restrict__ xyz, uint32_t M,
                                                     float* __restrict__ sigmas, float* __restrict__ rgbs) {
    const uint32_t i = blockIdx.x * NB + threadIdx.x;
    if (i >= M) return;
    const float3p q = reinterpret_cast<const float3p*>(xyz)[i];
    float s, c[3];
    decode_point<NL>(p, q.x, q.y, q.z, s, c);
    sigmas[i] = s;
    if (rgbs) reinterpret_cast<float3p*>(rgbs)[i] = float3p{c[0], c[1], c[2]};
}

template <int NL>
__global__ __launch_bounds__(NB) void k_render_rays(DecoderParams dp, MarchParams mp, const float* __restrict__ rays_o,
                                                    const float* __restrict__ rays_d, const float* __restrict__ aabb,
                                                    uint32_t N, float min_near, float T_thresh,
                                                    float* __restrict__ weights_sum, float* __restrict__ depth,
                                                    float* __restrict__ image, int32_t* __restrict__ n_samples) {
    const uint32_t n = blockIdx.x * NB + threadIdx.x;
    if (n >= N) return;
    const float3p o = reinterpret_cast<const float3p*>(rays_o)[n];
    const float3p d = reinterpret_cast<const float3p*>(rays_d)[n];
    float near, far;
    near_far_one(o, d, aabb, min_near, near, far);
    GridWalker w;
    w.init(mp, rays_o + 3ull * n, rays_d + 3ull * n);
    float ws = 0.f, dep = 0.f, r = 0.f, g = 0.f, b = 0.f;
    float t = near;
    t += w.step_len(t) * 0.0f;                        // eval mode: noise = 0 (kernel_march_rays with perturb=False)
    const uint32_t cnt = w.walk(t, far, mp.max_steps, [&](float cx, float cy, float cz, float tn, float dt) {
        float sigma, c[3];
        decode_point<NL>(dp, cx, cy, cz, sigma, c);
        const float alpha = 1.0f - __expf(-sigma * dt);         // kernel_composite_rays, raymarching.cu:877-899
        const float T = 1 - ws;
        const float wgt = alpha * T;
        ws += wgt;
        dep += wgt / tn;
        r += wgt * c[0];
        g += wgt * c[1];
        b += wgt * c[2];
        return !(T < T_thresh);
    });
    weights_sum[n] = ws;
    depth[n] = dep;
    reinterpret_cast<float3p*>(image)[n] = float3p{r, g, b};
    if (n_samples) n_samples[n] = (int32_t)cnt;
}

// rays for b views of h x w pixels: dir_cam = ((x+.5-cx)/fx, (y+.5-cy)/fy, 1); rays_d = normalize(R dir_cam); rays_o = t
__global__ __launch_bounds__(NB) void k_camera_rays(const float* __restrict__ intr, const float* __restrict__ poses, int nb, int h,
                                                    int w, float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                    float* __restrict__ dir_norm) {
    const size_t i = (size_t)blockIdx.x * NB + threadIdx.x;
    const size_t total = (size_t)nb * h * w;
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h), v = (int)(i / ((size_t)w * h));
    const float fx = intr[4 * v], fy = intr[4 * v + 1], cx = intr[4 * v + 2], cy = intr[4 * v + 3];
    const float* P = poses + 12 * v;                 // [3][4] row major
    const float dx = ((float)x + 0.5f - cx) / fx, dy = ((float)y + 0.5f - cy) / fy;
    float rx = dx * P[0] + dy * P[1] + P[2];
    float ry = dx * P[4] + dy * P[5] + P[6];
    float rz = dx * P[8] + dy * P[9] + P[10];
    const float nrm = fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);
    reinterpret_cast<float3p*>(rays_d)[i] = float3p{rx / nrm, ry / nrm, rz / nrm};
    reinterpret_cast<float3p*>(rays_o)[i] = float3p{P[3], P[7], P[11]};
    if (dir_norm) dir_norm[i] = sqrtf(dx * dx + dy * dy + 1.0f);
}

__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&c)[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void add_normalized(const float (&v)[3], float (&acc)[3]) {
    const float n = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    acc[0] += v[0] / n; acc[1] += v[1] / n; acc[2] += v[2] / n;
}

// BaseNeRF.render tail (base_nerf.py:543-554) fused: depth (1/z) and alpha -> normal_fg = depth_to_normal(depth/alpha),
// normal = normal_fg*alpha + normal_bg*(1-alpha).  alpha == nullptr: plain depth_to_normal.
__global__ __launch_bounds__(NB) void k_depth_to_normal(const float* __restrict__ depth, const float* __restrict__ alpha,
                                                        int alpha_stride, const float* __restrict__ intr, int nb, int h, int w,
                                                        float bg0, float bg1, float bg2, float* __restrict__ normal_fg,
                                                        float* __restrict__ normal) {
    const size_t i = (size_t)blockIdx.x * NB + threadIdx.x;
    const size_t total = (size_t)nb * h * w;
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h), v = (int)(i / ((size_t)w * h));
    const float fx = intr[4 * v], fy = intr[4 * v + 1], cx = intr[4 * v + 2], cy = intr[4 * v + 3];
    auto point = [&](int xx, int yy, float (&p)[3]) {
        const size_t j = ((size_t)v * h + yy) * w + xx;
        float dz = depth[j];
        if (alpha) dz = dz / fmaxf(alpha[j * alpha_stride], 1e-6f);
        const float inv = 1.0f / fmaxf(dz, 1e-6f);
        p[0] = (((float)xx + 0.5f - cx) / fx) * inv;
        p[1] = (((float)yy + 0.5f - cy) / fy) * inv;
        p[2] = inv;
    };
    // replicate-padded finite differences (geometry_utils.py:128-135)
    const int xr = x < w - 1 ? x : w - 2, xl = x > 0 ? x - 1 : 0, yd = y < h - 1 ? y : h - 2, yu = y > 0 ? y - 1 : 0;
    float a[3], b[3], right[3], left[3], up[3], down[3];
    point(xr + 1, y, a); point(xr, y, b);
    for (int k = 0; k < 3; ++k) right[k] = a[k] - b[k];
    point(xl + 1, y, a); point(xl, y, b);
    for (int k = 0; k < 3; ++k) left[k] = -(a[k] - b[k]);
    point(x, yu + 1, a); point(x, yu, b);
    for (int k = 0; k < 3; ++k) up[k] = -(a[k] - b[k]);
    point(x, yd + 1, a); point(x, yd, b);
    for (int k = 0; k < 3; ++k) down[k] = a[k] - b[k];
    float c[3], s[3] = {0.f, 0.f, 0.f};
    cross3(right, up, c); add_normalized(c, s);
    cross3(up, left, c); add_normalized(c, s);
    cross3(left, down, c); add_normalized(c, s);
    cross3(down, right, c); add_normalized(c, s);
    const float n = fmaxf(sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]), 1e-12f);
    const float nf[3] = {s[0] / n / 2 + 0.5f, -s[1] / n / 2 + 0.5f, -s[2] / n / 2 + 0.5f};       // to OpenGL, to [0,1]
    if (normal_fg) reinterpret_cast<float3p*>(normal_fg)[i] = float3p{nf[0], nf[1], nf[2]};
    if (normal) {
        const float al = alpha ? alpha[i * alpha_stride] : 1.0f;
        reinterpret_cast<float3p*>(normal)[i] = float3p{nf[0] * al + bg0 * (1 - al), nf[1] * al + bg1 * (1 - al), nf[2] * al + bg2 * (1 - al)};
    }
}

// normalize_depth (geometry_utils.py:151-168): one block per view for the two reductions, then elementwise
__global__ __launch_bounds__(NB) void k_normalize_depth(const float* __restrict__ depths, const float* __restrict__ alphas, int hw,
                                                        float far_depth, float alpha_clip, float eps, float* __restrict__ out) {
    __shared__ float smax[NB], smin[NB];
    const int v = blockIdx.x;
    const float* d = depths + (size_t)v * hw;
    const float* a = alphas + (size_t)v * hw;
    float mx = -FLT_MAX, mn = FLT_MAX;
    for (int i = threadIdx.x; i < hw; i += NB) {
        mx = fmaxf(mx, d[i]);
        const float fg = a[i] < alpha_clip ? 1.0f / eps : d[i] / fmaxf(a[i], eps);
        mn = fminf(mn, fg);
    }
    smax[threadIdx.x] = mx; smin[threadIdx.x] = mn;
    __syncthreads();
    for (int s = NB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
        }
        __syncthreads();
    }
    mx = smax[0]; mn = smin[0];
    const float den = fmaxf(mx - mn, eps);
    for (int i = threadIdx.x; i < hw; i += NB) {
        float fg = (d[i] / fmaxf(a[i], eps) - mn) / den;
        fg = fg * (1 - far_depth) + far_depth;
        out[(size_t)v * hw + i] = fminf(fmaxf(fg * a[i], 0.0f), 1.0f);
    }
}

int fill_decoder(DecoderParams& p, const float* table, int n_levels, const float* scales, const uint32_t* res, const uint32_t* off,
                 const uint32_t* size, const float* w1, const float* b1, const float* w2, const float* b2, int hidden, float bound,
                 float blob_density, float blob_radius, float sigmoid_saturation) {
    MVE_CHECK(table && scales && res && off && size && w1 && b1 && w2 && b2, MVE_ERR_ARG, "nerf decoder: null pointer");
    MVE_CHECK(n_levels == 12 || n_levels == 14 || n_levels == 16, MVE_ERR_ARG, "nerf decoder: n_levels must be 12, 14 or 16 (got %d)", n_levels);
    MVE_CHECK(hidden > 0 && hidden <= MAX_HIDDEN, MVE_ERR_ARG, "nerf decoder: hidden width must be <= %d", MAX_HIDDEN);
    p.table = table; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.hidden = hidden;
    p.bound = bound; p.blob_density = blob_density; p.blob_inv_2r2 = 1.0f / (2.0f * blob_radius * blob_radius);
    p.sat_scale = 1.0f + 2.0f * sigmoid_saturation; p.sat_shift = sigmoid_saturation;
    p.g.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        p.g.scale[l] = scales[l]; p.g.res[l] = res[l]; p.g.off[l] = off[l]; p.g.size[l] = size[l];
        MVE_CHECK(size[l] > 0, MVE_ERR_ARG, "nerf decoder: empty level %d", l);
    }
    return MVE_OK;
}

}  // namespace

extern "C" {

int mve_hashgrid_mlp_decode(const float* d_xyz, uint32_t M, const float* d_table, int n_levels, const float* level_scale,
                            const uint32_t* level_res, const uint32_t* level_offset, const uint32_t* level_size,
                            const float* d_w1, const float* d_b1, const float* d_w2, const float* d_b2, int hidden, float bound,
                            float blob_density, float blob_radius, float sigmoid_saturation, float* d_sigmas, float* d_rgbs,
                            void* stream) {
    if (M == 0) return MVE_OK;
    MVE_CHECK(d_xyz && d_sigmas, MVE_ERR_ARG, "hashgrid_mlp_decode: null pointer");
    DecoderParams p;
    int rc = fill_decoder(p, d_table, n_levels, level_scale, level_res, level_offset, level_size, d_w1, d_b1, d_w2, d_b2, hidden,
                          bound, blob_density, blob_radius, sigmoid_saturation);
    if (rc) return rc;
    const unsigned grid = mve_cdiv(M, NB);
    hipStream_t s = (hipStream_t)stream;
    if (hidden == HID) {
        if (n_levels == 12) k_point_decode2<12><<<grid, NB, 0, s>>>(p, d_xyz, M, d_sigmas, d_rgbs);
        else if (n_levels == 14) k_point_decode2<14><<<grid, NB, 0, s>>>(p, d_xyz, M, d_sigmas, d_rgbs);
        else k_point_decode2<16><<<grid, NB, 0, s>>>(p, d_xyz, M, d_sigmas, d_rgbs);
    } else if (n_levels == 12) k_point_decode<12><<<grid, NB, 0, s>>>(p, d_xyz, M, d_sigmas, d_rgbs);
    else if (n_levels == 14) k_point_decode<14><<<grid, NB, 0, s>>>(p, d_xyz, M, d_sigmas, d_rgbs);
    else k_point_decode<16><<<grid, NB, 0, s>>>(p, d_xyz, M, d_sigmas, d_rgbs);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

size_t mve_hashgrid_mlp_backward_workspace_bytes(uint32_t M, int n_levels) {
    const size_t nblk = (M + XTY_ROWS - 1) / XTY_ROWS;
    return sizeof(float) * ((size_t)M * (HID + HID + 2 * n_levels + 4) + nblk * (size_t)(HID * 2 * n_levels + HID + 4 * HID + 4) + 64);
}

int mve_hashgrid_mlp_backward(const float* d_xyz, uint32_t M, const float* d_table, int n_levels, const float* level_scale,
                              const uint32_t* level_res, const uint32_t* level_offset, const uint32_t* level_size, const float* d_w1,
                              const float* d_b1, const float* d_w2, const float* d_b2, int hidden, float bound, float blob_density,
                              float blob_radius, float sigmoid_saturation, const float* d_grad_sigma, const float* d_grad_rgb,
                              float* d_grad_table, float* d_grad_w1, float* d_grad_b1, float* d_grad_w2, float* d_grad_b2,
                              void* d_workspace, size_t workspace_bytes, void* stream) {
    MVE_CHECK(d_grad_table && d_grad_w1 && d_grad_b1 && d_grad_w2 && d_grad_b2, MVE_ERR_ARG, "hashgrid_mlp_backward: null gradient output");
    MVE_CHECK(hidden == HID, MVE_ERR_ARG, "hashgrid_mlp_backward: hidden width must be %d", HID);
    hipStream_t s = (hipStream_t)stream;
    const int nin = 2 * n_levels;
    if (M == 0) {
        MVE_HIP(hipMemsetAsync(d_grad_w1, 0, sizeof(float) * HID * nin, s));
        MVE_HIP(hipMemsetAsync(d_grad_b1, 0, sizeof(float) * HID, s));
        MVE_HIP(hipMemsetAsync(d_grad_w2, 0, sizeof(float) * 4 * HID, s));
        MVE_HIP(hipMemsetAsync(d_grad_b2, 0, sizeof(float) * 4, s));
        return MVE_OK;
    }
    MVE_CHECK(d_xyz && d_grad_sigma && d_workspace, MVE_ERR_ARG, "hashgrid_mlp_backward: null pointer");
    MVE_CHECK(workspace_bytes >= mve_hashgrid_mlp_backward_workspace_bytes(M, n_levels), MVE_ERR_NOMEM, "hashgrid_mlp_backward: workspace too small");
    DecoderParams p;
    int rc = fill_decoder(p, d_table, n_levels, level_scale, level_res, level_offset, level_size, d_w1, d_b1, d_w2, d_b2, hidden,
                          bound, blob_density, blob_radius, sigmoid_saturation);
    if (rc) return rc;
    float* ws = (float*)d_workspace;
    float* dh = ws; ws += (size_t)M * HID;
    float* hr = ws; ws += (size_t)M * HID;
    float* enc = ws; ws += (size_t)M * nin;
    float* dout = ws; ws += (size_t)M * 4;
    float* part = ws;
    const unsigned grid = mve_cdiv(M, NB);
    if (n_levels == 12) k_decode_backward<12><<<grid, NB, 0, s>>>(p, d_xyz, M, d_grad_sigma, d_grad_rgb, d_grad_table, dh, hr, enc, dout);
    else if (n_levels == 14) k_decode_backward<14><<<grid, NB, 0, s>>>(p, d_xyz, M, d_grad_sigma, d_grad_rgb, d_grad_table, dh, hr, enc, dout);
    else k_decode_backward<16><<<grid, NB, 0, s>>>(p, d_xyz, M, d_grad_sigma, d_grad_rgb, d_grad_table, dh, hr, enc, dout);
    MVE_LAUNCH_CHECK();
    const int nblk = (int)mve_cdiv(M, XTY_ROWS);
    // dW1 [64][2NL] = dh^T enc, db1 = colsum(dh)
    k_xty<<<nblk, 256, 0, s>>>(dh, HID, enc, nin, M, part);
    MVE_LAUNCH_CHECK();
    k_xty_reduce<<<mve_cdiv(HID * nin + HID, 256), 256, 0, s>>>(part, nblk, HID * nin + HID, HID * nin, d_grad_w1, d_grad_b1);
    MVE_LAUNCH_CHECK();
    // dW2 [4][64] = do^T relu(h), db2 = colsum(do)
    k_xty<<<nblk, 256, 0, s>>>(dout, 4, hr, HID, M, part);
    MVE_LAUNCH_CHECK();
    k_xty_reduce<<<mve_cdiv(4 * HID + 4, 256), 256, 0, s>>>(part, nblk, 4 * HID + 4, 4 * HID, d_grad_w2, d_grad_b2);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_adam_step(float* d_param, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                  float eps, int step, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(d_param && d_grad && d_exp_avg && d_exp_avg_sq && step >= 1, MVE_ERR_ARG, "adam_step: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    k_adam<<<mve_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(d_param, d_grad, d_exp_avg, d_exp_avg_sq, n, lr, beta1, beta2, eps, bc1, bc2);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_nerf_render_rays(const float* d_rays_o, const float* d_rays_d, uint32_t N, const uint8_t* d_bitfield, uint32_t grid_size,
                         const float* d_aabb, float bound, float min_near, float dt_gamma, uint32_t max_steps, float T_thresh,
                         const float* d_table, int n_levels, const float* level_scale, const uint32_t* level_res,
                         const uint32_t* level_offset, const uint32_t* level_size, const float* d_w1, const float* d_b1,
                         const float* d_w2, const float* d_b2, int hidden, float blob_density, float blob_radius,
                         float sigmoid_saturation, float* d_weights_sum, float* d_depth, float* d_image, int32_t* d_n_samples,
                         void* stream) {
    if (N == 0) return MVE_OK;
    MVE_CHECK(d_rays_o && d_rays_d && d_bitfield && d_aabb && d_weights_sum && d_depth && d_image, MVE_ERR_ARG,
              "nerf_render_rays: null pointer");
    MVE_CHECK(grid_size > 0 && grid_size <= 1024 && max_steps > 0, MVE_ERR_ARG, "nerf_render_rays: bad grid");
    DecoderParams p;
    int rc = fill_decoder(p, d_table, n_levels, level_scale, level_res, level_offset, level_size, d_w1, d_b1, d_w2, d_b2, hidden,
                          bound, blob_density, blob_radius, sigmoid_saturation);
    if (rc) return rc;
    MarchParams mp;
    mp.grid = d_bitfield; mp.bound = bound; mp.contract = 0; mp.dt_gamma = dt_gamma; mp.max_steps = max_steps; mp.C = 1; mp.H = grid_size;
    const unsigned grid = mve_cdiv(N, NB);
    hipStream_t s = (hipStream_t)stream;
#define GO(NL) k_render_rays<NL><<<grid, NB, 0, s>>>(p, mp, d_rays_o, d_rays_d, d_aabb, N, min_near, T_thresh, d_weights_sum, d_depth, d_image, d_n_samples)
#define GO2(NL) k_render_rays2<NL><<<grid, NB, 0, s>>>(p, mp, d_rays_o, d_rays_d, d_aabb, N, min_near, T_thresh, d_weights_sum, d_depth, d_image, d_n_samples)
    if (hidden == HID) {
        if (n_levels == 12) GO2(12);
        else if (n_levels == 14) GO2(14);
        else GO2(16);
    } else if (n_levels == 12) GO(12);
    else if (n_levels == 14) GO(14);
    else GO(16);
#undef GO
#undef GO2
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_camera_rays(const float* d_intrinsics, const float* d_poses, int n_views, int h, int w, float* d_rays_o, float* d_rays_d,
                    float* d_dir_norm, void* stream) {
    const size_t total = (size_t)n_views * h * w;
    if (total == 0) return MVE_OK;
    MVE_CHECK(d_intrinsics && d_poses && d_rays_o && d_rays_d, MVE_ERR_ARG, "camera_rays: null pointer");
    k_camera_rays<<<mve_cdiv(total, NB), NB, 0, (hipStream_t)stream>>>(d_intrinsics, d_poses, n_views, h, w, d_rays_o, d_rays_d, d_dir_norm);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_depth_to_normal(const float* d_depth, const float* d_alpha, int alpha_stride, const float* d_intrinsics, int n_views, int h,
                        int w, const float* normal_bg3, float* d_normal_fg, float* d_normal, void* stream) {
    const size_t total = (size_t)n_views * h * w;
    if (total == 0) return MVE_OK;
    MVE_CHECK(d_depth && d_intrinsics && (d_normal_fg || d_normal) && h >= 2 && w >= 2, MVE_ERR_ARG, "depth_to_normal: bad arguments");
    const float b0 = normal_bg3 ? normal_bg3[0] : 0.5f, b1 = normal_bg3 ? normal_bg3[1] : 0.5f, b2 = normal_bg3 ? normal_bg3[2] : 1.0f;
    k_depth_to_normal<<<mve_cdiv(total, NB), NB, 0, (hipStream_t)stream>>>(d_depth, d_alpha, alpha_stride > 0 ? alpha_stride : 1,
                                                                           d_intrinsics, n_views, h, w, b0, b1, b2, d_normal_fg, d_normal);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_normalize_depth(const float* d_depths, const float* d_alphas, int n_views, int hw, float far_depth, float alpha_clip, float eps,
                        float* d_out, void* stream) {
    if (n_views == 0 || hw == 0) return MVE_OK;
    MVE_CHECK(d_depths && d_alphas && d_out, MVE_ERR_ARG, "normalize_depth: null pointer");
    k_normalize_depth<<<n_views, NB, 0, (hipStream_t)stream>>>(d_depths, d_alphas, hw, far_depth, alpha_clip, eps, d_out);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
