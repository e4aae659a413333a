// Hardware probes for gfx950 that the attention kernel's design relies on (semantics of the cross-lane swaps and of the LDS transpose
// read; issue rates of the softmax instructions next to the MFMA shapes).  Standalone: hipcc --offload-arch=gfx950 -O3 probe.hip -o probe
// Not part of the product; output is kept under profiles/.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_swap(unsigned* out) {
    const unsigned l = threadIdx.x;
    auto a = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}

// mode 0: lane address = 8 * lane (contiguous 512 B); mode 1: the attention layout: row (lane >> 2) of 96-byte rows, 8-byte piece lane & 3
__global__ void k_tr(short* out, int mode) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const int byte = mode == 0 ? l * 8 : (l >> 2) * 96 + (l & 3) * 8;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + byte));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

// ---- issue-rate probes: N_IT iterations of 16 independent instructions per wave; cycles from s_memtime -------------------------------
constexpr int N_IT = 2048;
template <int OP>
__global__ __launch_bounds__(256) void k_rate(float* out, long long* cyc, float seed) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + 0.001f * (threadIdx.x + i);
    h8 ha, hb; h4 ha4, hb4;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * i + seed); hb[i] = (_Float16)(0.02f * i); }
    for (int i = 0; i < 4; ++i) { ha4[i] = ha[i]; hb4[i] = hb[i]; }
    f4 acc[8]; f16v acc32[2];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_IT; ++it) {
        if constexpr (OP == 0) { for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]); }
        if constexpr (OP == 1) { for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f); }
        if constexpr (OP == 2) { for (int i = 0; i < 16; i += 2) { f2 v = {x[i], x[i + 1]}; v = __builtin_elementwise_fma(v, f2{1.0001f, 1.0001f}, f2{0.5f, 0.5f}); x[i] = v[0]; x[i + 1] = v[1]; } }
        if constexpr (OP == 3) { for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], x[(i + 1) & 15]), x[(i + 2) & 15]) + 1.0f; }
        if constexpr (OP == 4) { for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0); }
        if constexpr (OP == 5) { for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(ha4, hb4, acc[i], 0, 0, 0); }
        if constexpr (OP == 6) { for (int i = 0; i < 2; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc32[i], 0, 0, 0); }
        if constexpr (OP == 7) {     // 8 MFMA 16x16x32 interleaved with 16 exp2: do they overlap inside one wave?
            for (int i = 0; i < 8; ++i) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0); x[2 * i] = __builtin_amdgcn_exp2f(x[2 * i]); x[2 * i + 1] = __builtin_amdgcn_exp2f(x[2 * i + 1]); }
        }
        if constexpr (OP == 8) {     // 8 MFMA + 32 plain fma
            for (int i = 0; i < 8; ++i) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0); for (int j = 0; j < 4; ++j) x[(4 * i + j) & 15] = __builtin_fmaf(x[(4 * i + j) & 15], 1.0001f, 0.5f); }
        }
        if constexpr (OP == 9) {     // packed convert f32 -> f16 pairs
            for (int i = 0; i < 16; i += 2) { auto h = __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]); x[i] += (float)h[0]; }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 2; ++i) s += acc32[i][0] + acc32[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void rate(const char* name, int per_iter, float* d_out, long long* d_cyc) {
    for (int bpc : {1, 2}) {        // 256-thread blocks per CU: 1 or 2 waves per SIMD
        const int grid = 256 * bpc;
        k_rate<OP><<<grid, 256>>>(d_out, d_cyc, 0.25f);
        CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        k_rate<OP><<<grid, 256>>>(d_out, d_cyc, 0.25f);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        std::vector<long long> c(grid);
        CK(hipMemcpy(c.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
        double m = 0; for (auto v : c) m += (double)v; m /= grid;
        // s_memtime ticks at a fixed 100 MHz-class clock on some parts: also report wall time per instruction per SIMD
        const double ns_per = ms * 1e6 / ((double)N_IT * per_iter * bpc);
        printf("%-34s waves/SIMD %d: %8.2f memtime-ticks/instr/wave   wall %7.3f ns per instr per SIMD (= %5.2f cyc @2.4GHz)\n", name, bpc,
               m / ((double)N_IT * per_iter), ns_per, ns_per * 2.4);
    }
}

int main() {
    unsigned* d_u; CK(hipMalloc(&d_u, 256 * 4));
    k_swap<<<1, 64>>>(d_u); CK(hipDeviceSynchronize());
    unsigned h[256]; CK(hipMemcpy(h, d_u, sizeof(h), hipMemcpyDeviceToHost));
    const char* names[4] = {"permlane16_swap r[0] (vdst=lane)", "permlane16_swap r[1] (src=100+lane)", "permlane32_swap r[0]", "permlane32_swap r[1]"};
    for (int k = 0; k < 4; ++k) { printf("%s:", names[k]); for (int l = 0; l < 64; l += 1) printf(" %u", h[64 * k + l]); printf("\n"); }
    short* d_s; CK(hipMalloc(&d_s, 256 * 2));
    for (int mode = 0; mode < 2; ++mode) {
        k_tr<<<1, 64>>>(d_s, mode); CK(hipDeviceSynchronize());
        short hs[256]; CK(hipMemcpy(hs, d_s, sizeof(hs), hipMemcpyDeviceToHost));
        printf("ds_read_b64_tr_b16 mode %d (lds[i] = i, 16-bit):\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d%s", l, hs[4 * l], hs[4 * l + 1], hs[4 * l + 2], hs[4 * l + 3], (l & 3) == 3 ? "\n" : ""); }
    }
    float* d_out; long long* d_cyc;
    CK(hipMalloc(&d_out, 512 * 256 * 4)); CK(hipMalloc(&d_cyc, 512 * 8));
    rate<0>("v_exp_f32", 16, d_out, d_cyc);
    rate<1>("v_fma_f32", 16, d_out, d_cyc);
    rate<2>("v_pk_fma_f32 (per packed instr)", 8, d_out, d_cyc);
    rate<3>("v_max3 + v_add (per pair)", 16, d_out, d_cyc);
    rate<4>("mfma 16x16x32 f16", 8, d_out, d_cyc);
    rate<5>("mfma 16x16x16 f16 (legacy)", 8, d_out, d_cyc);
    rate<6>("mfma 32x32x16 f16", 2, d_out, d_cyc);
    rate<7>("8 mfma16x16x32 + 16 exp (per iter/24)", 24, d_out, d_cyc);
    rate<8>("8 mfma16x16x32 + 32 fma (per iter/40)", 40, d_out, d_cyc);
    rate<9>("cvt_pkrtz + cvt + add (per triple)", 8, d_out, d_cyc);
    return 0;
}
