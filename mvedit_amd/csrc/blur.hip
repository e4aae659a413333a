// Separable Gaussian blur with reflect padding, its adjoint, and the reference's `highpass` built on it (gfx950; HBM-bound, two passes):
// torchvision's gaussian_blur as lib/pipelines/utils.py:187-188 and lib/pipelines/mvedit_3d_pipeline.py:473, :623-624, :671, :815-816 call
// it -- a 31 x 31 depthwise correlation there (961 taps per output), 2 x 31 taps here.  Arithmetic in blur_core.h (host/device).
#include "common.h"

#include "blur_core.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_K = 129;

struct Taps { float w[MAX_K]; };

// pass over rows (stride 1) or columns (stride W) of `planes` images; FINAL: out = base ? offset + base - blur : blur
template <bool ADJ>
__global__ __launch_bounds__(NT) void k_blur_pass(const float* __restrict__ x, int planes, int H, int W, int along_w, Taps taps, int r,
                                                  const float* __restrict__ base, float offset, float* __restrict__ out) {
    __shared__ float w[MAX_K];
    for (int t = threadIdx.x; t <= 2 * r; t += NT) w[t] = taps.w[t];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)planes * H * W) return;
    const int c = (int)(i % W), y = (int)((i / W) % H);
    const size_t p = i / ((size_t)H * W);
    const float* line = along_w ? x + p * H * W + (size_t)y * W : x + p * H * W + c;
    const int n = along_w ? W : H, stride = along_w ? 1 : W, pos = along_w ? c : y;
    const float b = ADJ ? bl_adj1d(line, n, stride, w, r, pos) : bl_fwd1d(line, n, stride, w, r, pos);
    out[i] = base ? offset + base[i] - b : b;
}

}  // namespace

extern "C" {

int mve_gaussian_blur(const float* d_x, int planes, int H, int W, int ksize, float sigma, int adjoint, const float* d_base, float offset,
                      float* d_tmp, float* d_out, void* stream) {
    if (planes == 0 || H == 0 || W == 0) return MVE_OK;
    MVE_CHECK(d_x && d_tmp && d_out && planes > 0 && H > 0 && W > 0, MVE_ERR_ARG, "gaussian_blur: bad arguments");
    MVE_CHECK(ksize >= 1 && ksize % 2 == 1 && ksize <= MAX_K && sigma > 0.f, MVE_ERR_ARG, "gaussian_blur: kernel size must be odd, <= %d, sigma > 0", MAX_K);
    MVE_CHECK(ksize / 2 < H && ksize / 2 < W, MVE_ERR_ARG, "gaussian_blur: reflect padding %d needs an image larger than that (%d x %d)", ksize / 2, H, W);
    MVE_CHECK(d_tmp != d_x && d_tmp != d_out, MVE_ERR_ARG, "gaussian_blur: tmp must not alias x or out");
    Taps taps;
    bl_kernel1d(ksize, sigma, taps.w);
    const size_t n = (size_t)planes * H * W;
    const unsigned grid = mve_cdiv(n, NT);
    hipStream_t s = (hipStream_t)stream;
    if (adjoint) {
        k_blur_pass<true><<<grid, NT, 0, s>>>(d_x, planes, H, W, 1, taps, ksize / 2, nullptr, 0.f, d_tmp);
        MVE_LAUNCH_CHECK();
        k_blur_pass<true><<<grid, NT, 0, s>>>(d_tmp, planes, H, W, 0, taps, ksize / 2, d_base, offset, d_out);
    } else {
        k_blur_pass<false><<<grid, NT, 0, s>>>(d_x, planes, H, W, 1, taps, ksize / 2, nullptr, 0.f, d_tmp);
        MVE_LAUNCH_CHECK();
        k_blur_pass<false><<<grid, NT, 0, s>>>(d_tmp, planes, H, W, 0, taps, ksize / 2, d_base, offset, d_out);
    }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
