// Fixed-order block reduction shared by the loss kernels (recon_loss.hip, mesh_reg.hip): a shared-memory tree whose result does not depend on
// the launch geometry beyond the block size, so per-block partial sums -- and the single-block pass over them -- are bitwise reproducible.
#pragma once
#include "common.h"

// every thread of the block must call it; all threads receive the sum.  sh: NT floats of shared memory.
template <int NT>
__device__ __forceinline__ float mve_block_sum(float v, float* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}
