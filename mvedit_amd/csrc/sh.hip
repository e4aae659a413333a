// Spherical-harmonics direction encoding (gfx950; one thread per point, registers only): the reference's lib/ops/shencoder
// (src/shencoder.cu:28-337 kernel_sh, :359-384 kernel_sh_backward) -- basis polynomials from recurrences instead of 256 hard-coded
// closed forms (sh_core.h).  fp32 like the reference's wrapper (custom_fwd(cast_inputs=torch.float32), sphere_harmonics.py:17).
#include "common.h"

#include "sh_core.h"

namespace {

constexpr int NT = 256;

struct ShK { float k[MVE_SH_MAX_DEGREE * MVE_SH_MAX_DEGREE]; };

__global__ __launch_bounds__(NT) void k_sh_encode(const float* __restrict__ inputs, uint32_t B, int C, ShK kk, float* __restrict__ outputs,
                                                  float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * NT + threadIdx.x;
    __shared__ float k[MVE_SH_MAX_DEGREE * MVE_SH_MAX_DEGREE];
    if (threadIdx.x < MVE_SH_MAX_DEGREE * MVE_SH_MAX_DEGREE) k[threadIdx.x] = kk.k[threadIdx.x];
    __syncthreads();
    if (b >= B) return;
    const int C2 = C * C;
    // every basis value is written exactly once, straight to its slot (no per-thread staging array: it would live in scratch)
    she_eval(k, inputs[3 * b], inputs[3 * b + 1], inputs[3 * b + 2], C, outputs + (size_t)b * C2, dy_dx ? dy_dx + (size_t)b * 3 * C2 : nullptr);
}

// grad_inputs[b][d] = sum_ch grad[b][ch] * dy_dx[b][d][ch]   (one thread per (point, axis), as the reference)
__global__ __launch_bounds__(NT) void k_sh_backward(const float* __restrict__ grad, const float* __restrict__ dy_dx, uint32_t B, int C2,
                                                    float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * NT + threadIdx.x;
    if (t >= B * 3) return;
    const uint32_t b = t / 3;
    const float* g = grad + (size_t)b * C2;
    const float* j = dy_dx + (size_t)t * C2;
    float acc = 0.f;
    for (int ch = 0; ch < C2; ++ch) acc += g[ch] * j[ch];
    grad_inputs[t] = acc;
}

}  // namespace

extern "C" {

int mve_sh_encode(const float* d_inputs, uint32_t B, int degree, float* d_outputs, float* d_dy_dx, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_inputs && d_outputs, MVE_ERR_ARG, "sh_encode: null pointer");
    MVE_CHECK(degree >= 1 && degree <= MVE_SH_MAX_DEGREE, MVE_ERR_ARG, "sh_encode: degree must be in [1, %d], got %d", MVE_SH_MAX_DEGREE, degree);
    ShK kk;
    she_constants(kk.k);
    k_sh_encode<<<mve_cdiv(B, NT), NT, 0, (hipStream_t)stream>>>(d_inputs, B, degree, kk, d_outputs, d_dy_dx);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_sh_encode_backward(const float* d_grad, const float* d_dy_dx, uint32_t B, int degree, float* d_grad_inputs, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(d_grad && d_dy_dx && d_grad_inputs, MVE_ERR_ARG, "sh_encode_backward: null pointer");
    MVE_CHECK(degree >= 1 && degree <= MVE_SH_MAX_DEGREE, MVE_ERR_ARG, "sh_encode_backward: degree must be in [1, %d], got %d", MVE_SH_MAX_DEGREE, degree);
    k_sh_backward<<<mve_cdiv((size_t)B * 3, NT), NT, 0, (hipStream_t)stream>>>(d_grad, d_dy_dx, B, degree * degree, d_grad_inputs);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
