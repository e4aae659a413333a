// Shared host/device helpers for libmvedit_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/mvedit_amd.h"

// ---------------------------------------------------------------------------
// error plumbing: every C-ABI entry returns 0 or a negative mve_status and
// leaves a message retrievable with mve_last_error().
// ---------------------------------------------------------------------------
void mve_set_error(const char* fmt, ...);

#define MVE_CHECK(cond, code, ...)                                     \
    do {                                                               \
        if (!(cond)) {                                                 \
            mve_set_error(__VA_ARGS__);                                \
            return (code);                                             \
        }                                                              \
    } while (0)

#define MVE_HIP(expr)                                                  \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) {                                        \
            mve_set_error("%s failed: %s (%s:%d)", #expr,              \
                          hipGetErrorString(_e), __FILE__, __LINE__);  \
            return MVE_ERR_HIP;                                        \
        }                                                              \
    } while (0)

#define MVE_LAUNCH_CHECK()                                             \
    do {                                                               \
        hipError_t _e = hipGetLastError();                             \
        if (_e != hipSuccess) {                                        \
            mve_set_error("kernel launch failed: %s (%s:%d)",          \
                          hipGetErrorString(_e), __FILE__, __LINE__);  \
            return MVE_ERR_HIP;                                        \
        }                                                              \
    } while (0)

static inline unsigned mve_cdiv(unsigned long long a, unsigned long long b) {
    return (unsigned)((a + b - 1) / b);
}

// ---------------------------------------------------------------------------
// 16-bit storage types.  Kernels are templated on a tag so fp16 and bf16 share
// one source; accumulation is always fp32.
// ---------------------------------------------------------------------------
typedef _Float16 f16;
typedef __bf16 bf16;

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct F16Tag {
    typedef f16 T;
    typedef f16x8 V8;
    static constexpr int dtype = MVE_F16;
    static __device__ __forceinline__ float to_f32(T v) { return (float)v; }
    static __device__ __forceinline__ T from_f32(float v) { return (T)v; }
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

struct BF16Tag {
    typedef bf16 T;
    typedef bf16x8 V8;
    static constexpr int dtype = MVE_BF16;
    static __device__ __forceinline__ float to_f32(T v) { return (float)v; }
    static __device__ __forceinline__ T from_f32(float v) { return (T)v; }
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------
// The low half of a residual-stream pair: 8 bits per element (round 5; 16 bits in round 4).  A stream value v is stored as
// hi = round16(v) -- the tensor every MFMA operand read sees -- plus lo8 = E5M2(2^8 (v - hi)) (v_cvt_pk_bf8_f32: round to nearest even; gfx950's
// bf8 is the OCP E5M2, what torch.float8_e5m2 holds).  |v - hi| <= ulp(hi) / 2, so two mantissa bits of it leave hi + lo within 2^-3 of half a step:
// ~3 more bits than hi alone, which is all the end-to-end budget needs (tests/rounding_budget_experiment.py --lo8: 8.6e-4 against 8.7e-4 with a
// 16-bit low half, 1.23e-3 without one) at half the low half's HBM traffic.  The scale keeps the remainder of any |v| >= 2^-11 out of E5M2's
// subnormals.  Four elements per dword, element e in byte e.
// ---------------------------------------------------------------------------
// Range: E5M2's largest finite value is 57344.  fp16 streams cannot reach it (the scaled remainder of the largest fp16 step is
// 2^12); a bf16 stream value of magnitude >= 2^16 has a remainder of up to 2^(e-8) and would overflow -- the scaled remainder is clamped to the
// format's range (the pair then reconstructs hi plus a saturated correction: never worse than hi alone), and a NaN remainder (hi non-finite)
// becomes 0 so that the pair reconstructs hi.
__device__ __forceinline__ float mve_lo8_scaled(float r) {
    const float s = __builtin_amdgcn_fmed3f(r * 256.f, -57344.f, 57344.f);
    return s == s ? s : 0.f;
}
// CLAMP = false (fp16 streams): the plain scale -- the instruction stream of round 5, bit for bit
template <bool CLAMP = false>
__device__ __forceinline__ unsigned mve_lo8_pack4(float a, float b, float c, float d) {
    int p;
    if constexpr (CLAMP) {
        p = __builtin_amdgcn_cvt_pk_bf8_f32(mve_lo8_scaled(a), mve_lo8_scaled(b), 0, false);
        p = __builtin_amdgcn_cvt_pk_bf8_f32(mve_lo8_scaled(c), mve_lo8_scaled(d), p, true);
    } else {
        p = __builtin_amdgcn_cvt_pk_bf8_f32(a * 256.f, b * 256.f, 0, false);
        p = __builtin_amdgcn_cvt_pk_bf8_f32(c * 256.f, d * 256.f, p, true);
    }
    return (unsigned)p;
}
__device__ __forceinline__ void mve_lo8_unpack4(unsigned p, float (&o)[4]) {
    const auto a = __builtin_amdgcn_cvt_pk_f32_bf8((int)p, false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)p, true);
    o[0] = a[0] * 0.00390625f; o[1] = a[1] * 0.00390625f; o[2] = b[0] * 0.00390625f; o[3] = b[1] * 0.00390625f;
}
// 8 consecutive elements: hi + lo (exact in fp32) from the 16-bit vector and the two lo8 dwords
template <class Tag>
__device__ __forceinline__ void mve_pair_load8(const typename Tag::V8& h, u32x2 l, float (&v)[8]) {
    float a[4], b[4];
    mve_lo8_unpack4(l[0], a);
    mve_lo8_unpack4(l[1], b);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = Tag::to_f32(h[e]) + a[e]; v[4 + e] = Tag::to_f32(h[4 + e]) + b[e]; }
}
// the pair of 8 fp32 values: hi (16-bit vector) and the two lo8 dwords
template <class Tag>
__device__ __forceinline__ u32x2 mve_pair_split8(const float (&v)[8], typename Tag::V8& hi) {
#pragma clang fp contract(off)
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = Tag::from_f32(v[e]); r[e] = v[e] - Tag::to_f32(hi[e]); }
    constexpr bool CL = Tag::dtype == MVE_BF16;      // only a bf16 remainder can leave E5M2's range
    return u32x2{mve_lo8_pack4<CL>(r[0], r[1], r[2], r[3]), mve_lo8_pack4<CL>(r[4], r[5], r[6], r[7])};
}

// dispatch a templated launcher on the runtime dtype code
#define MVE_DISPATCH_16(dtype, FN, ...)                                \
    ((dtype) == MVE_F16 ? FN<F16Tag>(__VA_ARGS__)                      \
     : (dtype) == MVE_BF16 ? FN<BF16Tag>(__VA_ARGS__)                  \
                           : (mve_set_error("unsupported dtype %d", (int)(dtype)), MVE_ERR_ARG))

// max over lanes {l, l^16} / {l, l^32} with gfx950's VALU half-row swaps (v_permlane16_swap / v_permlane32_swap)
// instead of ds_bpermute: no LDS round trip in the middle of a softmax dependency chain.
__device__ __forceinline__ float mve_max_xor16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float mve_max_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// XCD-aware bijective block remap (MI355X: 8 XCDs, block b lands on XCD b%8).
// Consecutive *logical* tiles share operand panels; give each XCD a contiguous
// chunk of the logical grid so those panels hit one L2.
__device__ __forceinline__ unsigned mve_xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned nx = 8u;
    if (nblk < nx * 2u) return bid;
    const unsigned q = nblk / nx, r = nblk % nx;
    const unsigned xcd = bid % nx, k = bid / nx;
    const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + k;
}
