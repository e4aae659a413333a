// What the chip sustains on the access pattern of the hash-grid renderer (csrc/nerf.hip, k_render_rays2): every lane of a wave fetches
// 8 bytes at an unrelated address of a table, G independent fetches in flight per lane, the result feeding a running sum.  Printed per
// table size (L2-resident, LLC-resident, HBM) and occupancy: lane-fetches per second and the bytes that would be if every fetch were a
// 64-byte sector.  The renderer's fetch rate (samples x 96 / time) is priced against these figures in bench.py / DESIGN.md 4.4.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/gather_probe tools/probe/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int G>
__global__ __launch_bounds__(256) void k_gather(const uint2* __restrict__ table, unsigned mask, int iters, unsigned long long* out) {
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint2 v[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            s = s * 1664525u + 1013904223u;
            v[g] = table[(s >> 7) & mask];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) acc += v[g].x ^ v[g].y;
        s ^= (unsigned)acc & 1u;                       // the next round's addresses wait for this round's data, as the trilinear blend does
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

template <int G>
double run(const uint2* table, unsigned mask, int blocks, int iters, unsigned long long* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_gather<G><<<blocks, 256>>>(table, mask, iters / 4, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_gather<G><<<blocks, 256>>>(table, mask, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)blocks * 256 * iters * G / (ms * 1e-3);
}

int main() {
    unsigned long long* out;
    CK(hipMalloc(&out, 64));
    const size_t sizes[] = {1u << 21, 24u << 20, 1u << 27, 1u << 30};      // 2 MiB, 24 MiB (one 4 MiB L2 per XCD cannot hold it), 128 MiB, 1 GiB
    for (size_t bytes : sizes) {
        uint2* table;
        size_t n = 1;
        while (n * 2 * sizeof(uint2) <= bytes) n *= 2;
        CK(hipMalloc(&table, n * sizeof(uint2)));
        CK(hipMemset(table, 1, n * sizeof(uint2)));
        for (int wpc : {8, 16, 32}) {                  // waves per CU = blocks of 4 waves x 256 CUs
            const int blocks = 256 * wpc / 4;          // ONE round of blocks: the dispatcher spreads them evenly, so wpc waves are resident per CU
            const double r1 = run<1>(table, (unsigned)(n - 1), blocks, 2048, out);
            const double r4 = run<4>(table, (unsigned)(n - 1), blocks, 1024, out);
            const double r8 = run<8>(table, (unsigned)(n - 1), blocks, 512, out);
            const double r16 = run<16>(table, (unsigned)(n - 1), blocks, 256, out);
            printf("table %7.1f MiB  %2d waves/CU resident: G=1 %6.1f  G=4 %6.1f  G=8 %6.1f  G=16 %6.1f  G lane-fetches/s   (x 64 B = %5.2f TB/s at G=16)\n",
                   n * sizeof(uint2) / 1048576.0, wpc, r1 / 1e9, r4 / 1e9, r8 / 1e9, r16 / 1e9, r16 * 64 / 1e12);
        }
        CK(hipFree(table));
    }
    return 0;
}
