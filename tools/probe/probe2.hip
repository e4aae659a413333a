// Does the matrix pipe of a SIMD run concurrently with its VALU?  (a) two waves per SIMD, one issuing only MFMAs, the other only VALU /
// transcendental instructions; (b) one wave per SIMD issuing k VALU instructions after every MFMA.  All loops are inline asm so that the
// compiler adds nothing.  hipcc --offload-arch=gfx950 -O3 probe2.hip -o probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int N_IT = 4096;

#define MFMA16(acc) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA32(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

// role 0: 8 MFMA16 per iteration; role 1: 32 v_fma; role 2: 16 v_exp + 16 v_fma; role 3: idle; role 4: 4 MFMA32 per iteration
template <int R>
__device__ __forceinline__ float run_role(float seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + 0.01f * i); b[i] = (_Float16)(0.02f * i); }
    f4 acc[8];
    f16v acc32[4];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed * 0.001f * (i + 1);
    const float c1 = 0.999f, c2 = 0.0001f;
    for (int it = 0; it < N_IT; ++it) {
        if constexpr (R == 0) { MFMA16(acc[0]); MFMA16(acc[1]); MFMA16(acc[2]); MFMA16(acc[3]); MFMA16(acc[4]); MFMA16(acc[5]); MFMA16(acc[6]); MFMA16(acc[7]); }
        if constexpr (R == 4) { MFMA32(acc32[0]); MFMA32(acc32[1]); MFMA32(acc32[2]); MFMA32(acc32[3]); }
        if constexpr (R == 1) {
#pragma unroll
            for (int k = 0; k < 2; ++k) { FMA(x[0]); FMA(x[1]); FMA(x[2]); FMA(x[3]); FMA(x[4]); FMA(x[5]); FMA(x[6]); FMA(x[7]); FMA(x[8]); FMA(x[9]); FMA(x[10]); FMA(x[11]); FMA(x[12]); FMA(x[13]); FMA(x[14]); FMA(x[15]); }
        }
        if constexpr (R == 2) {
            EXP(x[0]); FMA(x[8]); EXP(x[1]); FMA(x[9]); EXP(x[2]); FMA(x[10]); EXP(x[3]); FMA(x[11]); EXP(x[4]); FMA(x[12]); EXP(x[5]); FMA(x[13]); EXP(x[6]); FMA(x[14]); EXP(x[7]); FMA(x[15]);
            EXP(x[0]); FMA(x[8]); EXP(x[1]); FMA(x[9]); EXP(x[2]); FMA(x[10]); EXP(x[3]); FMA(x[11]); EXP(x[4]); FMA(x[12]); EXP(x[5]); FMA(x[13]); EXP(x[6]); FMA(x[14]); EXP(x[7]); FMA(x[15]);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    for (int i = 0; i < 4; ++i) s += acc32[i][0];
    return s;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k_pair(float* out, float seed) {      // waves 0-3 role RA, waves 4-7 (same SIMDs) role RB
    float s;
    if (threadIdx.x < 256) s = run_role<RA>(seed); else s = run_role<RB>(seed);
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// one wave per SIMD: K VALU (fma) after every MFMA16 / MFMA32
template <int K, int BIG>
__global__ __launch_bounds__(256) void k_mix(float* out, float seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + 0.01f * i); b[i] = (_Float16)(0.02f * i); }
    f4 acc[4];
    f16v acc32[4];
    for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed * 0.001f * (i + 1);
    const float c1 = 0.999f, c2 = 0.0001f;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (BIG) MFMA32(acc32[m]); else MFMA16(acc[m]);
#pragma unroll
            for (int k = 0; k < K; ++k) FMA(x[(m * 4 + k) & 15]);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc32[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F> float timed(F f) {
    f(); CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}

int main() {
    float* d; CK(hipMalloc(&d, 256 * 512 * 4));
#define PAIR(RA, RB, name) printf("%-52s %8.3f ms\n", name, timed([&] { k_pair<RA, RB><<<256, 512>>>(d, 0.5f); }))
    PAIR(0, 3, "MFMA16 wave alone (8/iter)");
    PAIR(4, 3, "MFMA32 wave alone (4/iter)");
    PAIR(1, 3, "v_fma wave alone (32/iter)");
    PAIR(2, 3, "exp+fma wave alone (16+16/iter)");
    PAIR(0, 0, "MFMA16 + MFMA16 on one SIMD");
    PAIR(1, 1, "v_fma + v_fma on one SIMD");
    PAIR(2, 2, "exp+fma + exp+fma on one SIMD");
    PAIR(0, 1, "MFMA16 wave + v_fma wave on one SIMD");
    PAIR(0, 2, "MFMA16 wave + exp+fma wave on one SIMD");
    PAIR(4, 1, "MFMA32 wave + v_fma wave on one SIMD");
    PAIR(4, 2, "MFMA32 wave + exp+fma wave on one SIMD");
#define MIX(K, BIG) printf("one wave/SIMD: MFMA%s + %d v_fma each: %8.3f ms (4 MFMA per iter)\n", BIG ? "32" : "16", K, timed([&] { k_mix<K, BIG><<<256, 256>>>(d, 0.5f); }))
    MIX(0, 0); MIX(1, 0); MIX(2, 0); MIX(3, 0); MIX(4, 0); MIX(6, 0); MIX(8, 0);
    MIX(0, 1); MIX(2, 1); MIX(4, 1); MIX(6, 1); MIX(8, 1); MIX(12, 1);
    return 0;
}
