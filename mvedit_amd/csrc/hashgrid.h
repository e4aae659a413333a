// tiny-cuda-nn HashGrid level table, the decoder parameter block and the Smoothstep hash-grid encoding shared by nerf.hip (iNGPDecoder,
// lib/models/decoders/ingp_decoder.py:56-120) and triplane.hip (TriPlaneiNGPDecoder, lib/models/decoders/triplane_ingp_decoder.py:102-114).
#pragma once
#include "raymarch_core.h"

namespace {

constexpr int MAX_LEVELS = 16;
constexpr int MAX_HIDDEN = 64;

struct HashGridMeta {
    float scale[MAX_LEVELS];
    uint32_t res[MAX_LEVELS], off[MAX_LEVELS], size[MAX_LEVELS];
    int n_levels;
};

struct DecoderParams {
    const float* table;      // [rows][2]
    const float* w1;         // [hidden][2*n_levels]
    const float* b1;         // [hidden]
    const float* w2;         // [4][hidden]
    const float* b2;         // [4]
    int hidden;
    float bound, blob_density, blob_inv_2r2, sat_scale, sat_shift;
    HashGridMeta g;
};

// tiny-cuda-nn HashGrid, Smoothstep interpolation, 2 features per level (see oracle/nerf_oracle.py for the restated spec)
template <int NL>
__device__ __forceinline__ void hash_encode(const DecoderParams& p, float x, float y, float z, float (&enc)[2 * NL]) {
    const float inv = 1.0f / (2.0f * p.bound);
    const float u[3] = {(x + p.bound) * inv, (y + p.bound) * inv, (z + p.bound) * inv};
#pragma unroll
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
        // dense strides while they fit, else the coherent prime hash (grid_index of tiny-cuda-nn)
        const bool s1 = res <= size;                                  // stride after x
        const bool s2 = s1 && (uint64_t)res * res <= size;            // stride after y
        const uint64_t stride3 = (uint64_t)res * res * (s2 ? res : 1u);
        const bool hashed = s2 ? (size < stride3) : true;
        const bool pow2 = (size & (size - 1u)) == 0u;
        const float2p* tab = reinterpret_cast<const float2p*>(p.table) + p.g.off[l];
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            float wt = 1.0f;
            uint32_t c[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (corner & (1 << d)) { wt = wt * w[d]; c[d] = cell[d] + 1u; }
                else { wt = wt * (1.0f - w[d]); c[d] = cell[d]; }
            }
            // idx mod size without the 32-bit division (it was ~1/3 of the kernel's VALU work): hashed tables have 2^k rows;
            // a dense index of an in-range point is below 2*size, anything else (points outside the box) takes the slow path
            uint32_t idx;
            if (hashed) {
                idx = (c[0] * 1u) ^ (c[1] * 2654435761u) ^ (c[2] * 805459861u);
                idx = pow2 ? (idx & (size - 1u)) : (idx % size);
            } else {
                idx = c[0] + c[1] * res + c[2] * res * res;
                if (idx >= size) idx -= size;
                if (idx >= size) idx %= size;
            }
            const float2p f = tab[idx];
            a0 = fmaf(wt, f.x, a0);
            a1 = fmaf(wt, f.y, a1);
        }
        enc[2 * l] = a0;
        enc[2 * l + 1] = a1;
        // one level at a time: without this the scheduler hoists all 8*NL gathers to the top, needs > 256 VGPRs and the kernel
        // runs at one wave per SIMD -- latency-bound on its own dependent index arithmetic
        __builtin_amdgcn_sched_barrier(0);
    }
}

// transpose of hash_encode: g_table[row(corner)] += corner weight * denc[2 l .. 2 l + 1] (float atomics, as tiny-cuda-nn's backward)
template <int NL>
__device__ __forceinline__ void hash_scatter(const DecoderParams& p, float x, float y, float z, const float (&denc)[2 * NL], float* __restrict__ g_table) {
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
        float* gt = g_table + 2ull * p.g.off[l];
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
            atomicAdd(gt + 2ull * idx, wt * denc[2 * l]);
            atomicAdd(gt + 2ull * idx + 1, wt * denc[2 * l + 1]);
        }
    }
}

}  // namespace
