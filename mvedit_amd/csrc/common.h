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
