// Epilogue of the 256-row GEMM / conv tiles (gemm_big.hip: 256 x {320,256} x 64 two-stage kernel; gemm_pp.hip: the ping-pong main loop
// over the same tile).  A wave (wm, wn) of the 2 x 4 arrangement holds acc[j][i] = 16 x 16 fragment (i along M, j along N) of its
// 128 x BN2/4 tile in the swapped-operand layout (a lane owns 4 consecutive n of one row).  `smem` is the kernel's whole dynamic LDS
// allocation: every wave must have finished reading it (and all LDS-DMA into it must have landed) before the call.
#pragma once
#include "gemm_shared.h"

namespace {

template <class Tag, int BN2>
__device__ __forceinline__ void big_tile_epilogue(const GemmParams& p, f32x4 (&acc)[BN2 / 64][8], unsigned char* smem, int m0, int n0,
                                                  int kslice, int tid, int lane, int wm, int wn) {
    typedef typename Tag::V8 V8;
    typedef typename Tag::T T;
    constexpr int BM2 = 256, NTH = 512, MF = 8, WTM = 128;
    constexpr int WTN = BN2 / 4, NF = WTN / 16;
    constexpr int CS_LD = BN2 + 4;
    // ---- GEGLU epilogue: out[m][i] = (v[2i] + b[2i]) * gelu(v[2i+1] + b[2i+1]).  A lane owns 4 consecutive n = two
    // (value, gate) pairs of one row, so the activation is evaluated on the accumulators; only the 16-bit results (half the
    // columns) pass through LDS, in ONE 256-row pass, for 16-byte row-segment stores.  Same arithmetic as gemm_epilogue_store.
    if (p.geglu && p.splitk <= 1) {
        constexpr int HS_LD = BN2 / 2 + 8;              // 168 elements: conflict-free 4-byte writes (row stride 84 words)
        T* Hs = reinterpret_cast<T*>(smem);
        typedef T T2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int c = wn * WTN + j * 16 + (lane >> 4) * 4;
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n0 + c < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const f32x4 v = acc[j][i];
                T2 o;
                o[0] = Tag::from_f32((v[0] + b4[0]) * gelu_erf(v[1] + b4[1]));
                o[1] = Tag::from_f32((v[2] + b4[2]) * gelu_erf(v[3] + b4[3]));
                *reinterpret_cast<T2*>(Hs + (wm * WTM + i * 16 + (lane & 15)) * HS_LD + (c >> 1)) = o;
            }
        }
        __syncthreads();
        constexpr int HCH = BN2 / 16;                   // 20 chunks of 8 output columns per row
        for (int task = tid; task < BM2 * HCH; task += NTH) {
            const int r = task / HCH, ch = task - r * HCH;
            const int m = m0 + r, n = n0 + ch * 16;     // n: first of the 16 input columns behind this output chunk
            if (m >= p.M || n >= p.N) continue;
            *reinterpret_cast<V8*>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldc + (n >> 1)) = *reinterpret_cast<const V8*>(Hs + r * HS_LD + ch * 8);
        }
        return;
    }

    // ---- epilogue: four 64-row passes through an fp32 LDS tile, 16-byte row-segment stores ----------------------------
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CHUNKS = BN2 / 8;
    constexpr int TASKS = 64 * CHUNKS;
#pragma unroll
    for (int pass = 0; pass < BM2 / 64; ++pass) {
        if (wm == pass / 2) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    constexpr int dummy = 0; (void)dummy;
                    const int i = (pass & 1) * 4 + i4;
                    const int r = i4 * 16 + (lane & 15);
                    const int c = wn * WTN + j * 16 + (lane >> 4) * 4;
                    *reinterpret_cast<f32x4*>(Cs + r * CS_LD + c) = acc[j][i];
                }
        }
        __syncthreads();
        for (int task = tid; task < TASKS; task += NTH) {
            const int r = task / CHUNKS, ch = task - r * CHUNKS;
            const int m = m0 + pass * 64 + r, n = n0 + ch * 8;
            if (m >= p.M || n >= p.N) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + r * CS_LD + ch * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + r * CS_LD + ch * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
            if (p.splitk > 1) {
                float* pp = p.partial + ((size_t)kslice * p.M + m) * p.N + n;
                *reinterpret_cast<f32x4*>(pp) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(pp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                continue;
            }
            gemm_epilogue_store<Tag>(p, m, n, v);
        }
        __syncthreads();
    }
}

}  // namespace
