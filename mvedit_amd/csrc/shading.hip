// Per-pixel shading of rendered views (gfx950; HBM-bound, one pass): the reference's tone-mapping look-up table
// (lib/models/decoders/tonemapping.py:33-54) and the shading arithmetic its pipelines apply to every rendered batch
// (lib/pipelines/mvedit_3d_pipeline.py:1372-1384, same expression at :155-168) as single launches instead of ~25 elementwise torch
// kernels over [b, S, S, 3] tensors.  Compiled without fma contraction (build.py) so that the table interpolation evaluates the
// reference's expression op by op.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_STEPS = 64;

struct Lut {
    const float* x; const float* y; int n;
};

// torch.bucketize(v, table, right=True).clamp(1, n - 1): number of table entries <= v
__device__ __forceinline__ int bucket(const float* __restrict__ t, int n, float v) {
    int i = 0;
    for (int k = 0; k < n; ++k) i += (t[k] <= v) ? 1 : 0;
    return i < 1 ? 1 : (i > n - 1 ? n - 1 : i);
}

// piecewise-linear map from table a to table b (lut: a = lut_x, b = lut_y; inverse_lut: a = lut_y, b = lut_x)
__device__ __forceinline__ float interp(const float* __restrict__ a, const float* __restrict__ b, int n, float v) {
    const int i = bucket(a, n, v);
    const float t = (v - a[i - 1]) / (a[i] - a[i - 1]);
    return b[i - 1] + (b[i] - b[i - 1]) * t;
}

__global__ __launch_bounds__(NT) void k_lut(const float* __restrict__ v, size_t n, Lut l, int inverse, int linear, float* __restrict__ out) {
    __shared__ float tx[MAX_STEPS], ty[MAX_STEPS];
    if (threadIdx.x < l.n) { tx[threadIdx.x] = l.x[threadIdx.x]; ty[threadIdx.x] = l.y[threadIdx.x]; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    float x = v[i];
    if (!inverse) {
        if (linear) x = log2f(fmaxf(x, 1e-6f));
        out[i] = interp(tx, ty, l.n, x);
    } else {
        const float r = interp(ty, tx, l.n, x);
        out[i] = linear ? exp2f(r) : r;
    }
}

// image = lut(inverse_lut(rgb / max(alpha, 1e-6)) + log2(max(shading, 1e-6))) * alpha + bg * (1 - alpha)      (tone-mapped), or
// image = rgb * shading + bg * (1 - alpha)                                                                    (no table),
// shading = max(light . n_cv, 0) * (1 - ambient) + ambient,  n_cv = (2 n0 - 1, 1 - 2 n1, 1 - 2 n2)
__global__ __launch_bounds__(NT) void k_shade_views(const float* __restrict__ rgba, const float* __restrict__ normal_fg,
                                                    const float* __restrict__ lights, unsigned n_views, unsigned pix, float ambient,
                                                    float bg, Lut l, float* __restrict__ image) {
    __shared__ float tx[MAX_STEPS], ty[MAX_STEPS];
    if (l.n > 0 && threadIdx.x < l.n) { tx[threadIdx.x] = l.x[threadIdx.x]; ty[threadIdx.x] = l.y[threadIdx.x]; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)n_views * pix) return;
    const unsigned v = (unsigned)(i / pix);
    const f32x4 c = *reinterpret_cast<const f32x4*>(rgba + i * 4);
    const float n0 = normal_fg[i * 3] * 2.0f - 1.0f, n1 = -normal_fg[i * 3 + 1] * 2.0f + 1.0f, n2 = -normal_fg[i * 3 + 2] * 2.0f + 1.0f;
    const float dot = (lights[v * 3] * n0 + lights[v * 3 + 1] * n1) + lights[v * 3 + 2] * n2;
    const float shading = fmaxf(dot, 0.0f) * (1.0f - ambient) + ambient;
    const float a = c[3], back = bg * (1.0f - a);
    float o[3];
    if (l.n > 0) {
        const float ls = log2f(fmaxf(shading, 1e-6f)), den = fmaxf(a, 1e-6f);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = interp(tx, ty, l.n, interp(ty, tx, l.n, c[k] / den) + ls) * a + back;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = c[k] * shading + back;
    }
    image[i * 3] = o[0]; image[i * 3 + 1] = o[1]; image[i * 3 + 2] = o[2];
}

}  // namespace

extern "C" {

int mve_tonemap_lut(const float* d_x, size_t n, const float* d_lut_x, const float* d_lut_y, int steps, int inverse, int linear,
                    float* d_out, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(d_x && d_out && d_lut_x && d_lut_y && steps >= 2 && steps <= MAX_STEPS, MVE_ERR_ARG, "tonemap_lut: bad arguments (2 <= steps <= %d)", MAX_STEPS);
    k_lut<<<mve_cdiv(n, NT), NT, 0, (hipStream_t)stream>>>(d_x, n, Lut{d_lut_x, d_lut_y, steps}, inverse ? 1 : 0, linear ? 1 : 0, d_out);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_shade_views(const float* d_rgba, const float* d_normal_fg, const float* d_cam_lights, uint32_t n_views, uint32_t pix_per_view,
                    float ambient_light, float bg_color, const float* d_lut_x, const float* d_lut_y, int steps, float* d_image,
                    void* stream) {
    if (n_views == 0 || pix_per_view == 0) return MVE_OK;
    MVE_CHECK(d_rgba && d_normal_fg && d_cam_lights && d_image, MVE_ERR_ARG, "shade_views: null pointer");
    MVE_CHECK((d_lut_x == nullptr) == (d_lut_y == nullptr) && (d_lut_x == nullptr || (steps >= 2 && steps <= MAX_STEPS)), MVE_ERR_ARG,
              "shade_views: pass both tables (2 <= steps <= %d) or neither", MAX_STEPS);
    const size_t n = (size_t)n_views * pix_per_view;
    k_shade_views<<<mve_cdiv(n, NT), NT, 0, (hipStream_t)stream>>>(d_rgba, d_normal_fg, d_cam_lights, n_views, pix_per_view, ambient_light,
                                                                  bg_color, Lut{d_lut_x, d_lut_y, d_lut_x ? steps : 0}, d_image);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
