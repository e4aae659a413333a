// Exclusive scan of n values (device-wide, three launches), shared by the kernels whose output layout is data dependent
// (dmtet.hip, mesh_reg.hip).  Lives in an anonymous namespace: every translation unit gets its own instantiations.
#pragma once
#include "common.h"

namespace {

constexpr int SCAN_DB = 256;
constexpr int SCAN_ITEMS = 8;                  // elements per thread in the scan kernels
constexpr int SCAN_TILE = SCAN_DB * SCAN_ITEMS;

// ---- exclusive scan of n values of type T: tile sums -> single-block scan of the sums -> per-tile scan + base -------------
template <typename T>
__global__ __launch_bounds__(SCAN_DB) void k_tile_sums(const T* __restrict__ in, size_t n, T* __restrict__ sums) {
    __shared__ T red[SCAN_DB / 64];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    T v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) v += (base + k < n) ? in[base + k] : (T)0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { T t = 0; for (int k = 0; k < SCAN_DB / 64; ++k) t += red[k]; sums[blockIdx.x] = t; }
}
template <typename T>
__global__ __launch_bounds__(1024) void k_scan_sums(T* __restrict__ sums, size_t ntiles, T* __restrict__ total) {
    __shared__ T buf[1024];
    __shared__ T carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t base = 0; base < ntiles; base += 1024) {
        const size_t i = base + threadIdx.x;
        const T v = i < ntiles ? sums[i] : (T)0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const T add = threadIdx.x >= (unsigned)o ? buf[threadIdx.x - o] : (T)0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < ntiles) sums[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
template <typename T>
__global__ __launch_bounds__(SCAN_DB) void k_tile_scan(const T* __restrict__ in, size_t n, const T* __restrict__ sums, T* __restrict__ out) {
    __shared__ T wsum[SCAN_DB / 64];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    T loc[SCAN_ITEMS], t = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { loc[k] = (base + k < n) ? in[base + k] : (T)0; t += loc[k]; }
    T incl = t;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { const T u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    T pre = sums[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pre += wsum[w];
    T run = pre + incl - t;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = run; run += loc[k]; }
}
template <typename T>
int exclusive_scan(const T* in, T* out, size_t n, T* tile_sums, T* total, hipStream_t s) {
    const size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    k_tile_sums<T><<<(unsigned)ntiles, SCAN_DB, 0, s>>>(in, n, tile_sums);
    k_scan_sums<T><<<1, 1024, 0, s>>>(tile_sums, ntiles, total);
    k_tile_scan<T><<<(unsigned)ntiles, SCAN_DB, 0, s>>>(in, n, tile_sums, out);
    return hipGetLastError() == hipSuccess ? MVE_OK : MVE_ERR_HIP;
}

}  // namespace
