// Float-atomic scatter rate of the hash-grid backward's access pattern (csrc/nerf.hip k_decode_backward: per sample 12 levels x 8 corners x 2
// consecutive floats into a [rows][2] fp32 gradient table of up to 48 MiB; coarse levels hit a few thousand rows, fine levels are hashed):
//   device  : atomicAdd at agent scope into ONE table (what the kernel did through round 3);
//   xcd     : workgroup-scope atomics into a table private to the block's XCD (HW_REG_XCC_ID), 8 tables reduced afterwards -- the XCDs' L2s are
//             not coherent with each other, so agent-scope atomics cannot be served by an L2; private tables can.
// Prints M atomic pairs / s for both, for a small (coarse level) and a large (hashed level) table.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/atomic_probe tools/probe/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }   // HW_REG_XCC_ID[3:0]

template <int MODE>
__global__ __launch_bounds__(256) void k_scatter(float* __restrict__ table, unsigned rows_mask, size_t stride, int per_thread, int coherent_run) {
    float* t = table;
    if (MODE == 1) t += (size_t)xcc_id() * stride;
    unsigned s = (blockIdx.x * 256u + threadIdx.x) / (unsigned)coherent_run * 2654435761u + 12345u;     // `coherent_run` neighbouring samples share their rows
    for (int it = 0; it < per_thread; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned row = (s >> 7) & rows_mask;
        const float v = 1.0f + (float)(threadIdx.x & 3);
        if (MODE == 0) {
            atomicAdd(t + 2ull * row, v);
            atomicAdd(t + 2ull * row + 1, v);
        } else {
            __hip_atomic_fetch_add(t + 2ull * row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(t + 2ull * row + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

__global__ void k_reduce(const float* __restrict__ priv, float* __restrict__ out, size_t n, size_t stride) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int x = 0; x < 8; ++x) a += priv[x * stride + i];
    out[i] = a;
}

int main() {
    const size_t rows_big = 12ull << 19;                 // 12 levels x 2^19 rows = 48 MiB of [row][2] floats
    float *one, *priv, *red;
    CK(hipMalloc(&one, rows_big * 8)); CK(hipMalloc(&priv, 8 * rows_big * 8)); CK(hipMalloc(&red, rows_big * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 948, per = 96;                     // 242 k "samples" x 96 corner pairs = 23.3 M atomic pairs, as one nerf_optim iteration
    for (int run : {1, 8}) {
        for (unsigned rows : {4096u, 65536u, 1u << 19, (unsigned)rows_big / 2 + (unsigned)rows_big / 4}) {      // (masks: powers of two below)
            unsigned mask = 1;
            while (mask * 2 <= rows) mask *= 2;
            mask -= 1;
            for (int mode = 0; mode < 2; ++mode) {
                CK(hipMemset(one, 0, rows_big * 8)); CK(hipMemset(priv, 0, 8 * rows_big * 8));
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) k_scatter<0><<<blocks, 256>>>(one, mask, rows_big * 2, per, run);
                    else k_scatter<1><<<blocks, 256>>>(priv, mask, rows_big * 2, per, run);
                    CK(hipEventRecord(e1));
                    CK(hipDeviceSynchronize());
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                double chk = 0;
                if (mode == 1) { k_reduce<<<(unsigned)((rows_big * 2 + 255) / 256), 256>>>(priv, red, rows_big * 2, rows_big * 2); CK(hipDeviceSynchronize()); }
                float h[2];
                CK(hipMemcpy(h, mode ? red : one, 8, hipMemcpyDeviceToHost));
                chk = h[0];
                const double pairs = (double)blocks * 256 * per;
                printf("rows %8u  neighbours sharing rows %d  %-7s %8.3f ms  %7.1f M pairs/s   (table[0] = %.0f)\n", mask + 1, run, mode ? "xcd" : "device", ms,
                       pairs / ms / 1e3, chk);
            }
        }
    }
    float ms = 0;
    CK(hipEventRecord(e0));
    k_reduce<<<(unsigned)((rows_big * 2 + 255) / 256), 256>>>(priv, red, rows_big * 2, rows_big * 2);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("reduce of 8 private 48 MiB tables: %.3f ms\n", ms);
    return 0;
}
