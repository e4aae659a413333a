// LayerNorm row arithmetic shared by k_layernorm (csrc/norm.hip) and the GEMM epilogue that normalises the rows it has just produced
// (csrc/gemm_big_epilogue.h, LNF instantiation; round 6).  ONE definition with explicit fused multiply-adds, so that both callers round
// identically whatever hipcc's contraction choices are in their translation units: a row normalised inside the producing launch is bit-identical
// to the same row normalised by the stand-alone kernel (batch / tile invariance: which of the two runs depends on the launch geometry).
// Row layout: lane l of a wave holds chunk l, l + 64, ... (8 consecutive channels each); sums go lane-sequentially over a chunk, then over the
// wave by the xor-shuffle tree 32 .. 1; biased variance about the mean (two passes over registers), eps inside the square root.
#pragma once
#include "common.h"

namespace {

// s[l] + s[l ^ d] for d = 32, 16, 8, 4, 2, 1 -- the xor tree -- on the VALU cross-lane paths instead of six ds_bpermute round trips (~100 cycles of
// latency each: a row's two reductions were a 1 200-cycle dependent chain).  d = 32 / 16: v_permlane32_swap / v_permlane16_swap; d = 8: DPP row_ror:8
// (a rotation by 8 inside a 16-lane row IS xor 8); d = 4: row_ror:4 -- lane l reads lane (l + 4) mod 16, which is l ^ 4 or (l ^ 4) ^ 8, and after the
// d = 8 step those two hold the same value; d = 2, 1: quad_perm.  Every lane adds the same two operands as with __shfl_xor: identical bits.
template <int CTRL>
__device__ __forceinline__ float mve_ln_dpp_add(float s) {
    return s + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float mve_ln_wave_sum(float s) {
    { const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false); s = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false); s = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    s = mve_ln_dpp_add<0x128>(s);      // row_ror:8
    s = mve_ln_dpp_add<0x124>(s);      // row_ror:4
    s = mve_ln_dpp_add<0x4E>(s);       // quad_perm [2, 3, 0, 1]
    s = mve_ln_dpp_add<0xB1>(s);       // quad_perm [1, 0, 3, 2]
    return s;
}
// the same tree over R independent values
template <int R>
__device__ __forceinline__ void mve_ln_wave_sum_n(float (&s)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = mve_ln_wave_sum(s[r]);
}
__device__ __forceinline__ float mve_ln_sum8(const float (&v)[8], float s) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    return s;
}
__device__ __forceinline__ float mve_ln_sq8(const float (&v)[8], float mean, float q) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q = __builtin_fmaf(d, d, q); }
    return q;
}
template <class Tag>
__device__ __forceinline__ typename Tag::V8 mve_ln_out8(const float (&v)[8], float mean, float rstd, const float* __restrict__ gamma8, const float* __restrict__ beta8) {
    typename Tag::V8 pk;
#pragma unroll
    for (int e = 0; e < 8; ++e) pk[e] = Tag::from_f32(__builtin_fmaf((v[e] - mean) * rstd, gamma8[e], beta8[e]));
    return pk;
}

}  // namespace
