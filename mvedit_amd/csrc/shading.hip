// Per-pixel shading of rendered views (gfx950; HBM-bound, one pass): the reference's tone-mapping look-up table
// (lib/models/decoders/tonemapping.py:33-54) and the shading arithmetic its pipelines apply to every rendered batch
// (lib/pipelines/mvedit_3d_pipeline.py:1372-1384, same expression at :155-168) as single launches instead of ~25 elementwise torch
// kernels over [b, S, S, 3] tensors.  Compiled without fma contraction (build.py) so that the table interpolation evaluates the
// reference's expression op by op.
#include "common.h"

#include "shading_core.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_STEPS = 64;

struct Lut {
    const float* x; const float* y; int n;
};

__global__ __launch_bounds__(NT) void k_lut(const float* __restrict__ v, size_t n, Lut l, int inverse, int linear, float* __restrict__ out) {
    __shared__ float tx[MAX_STEPS], ty[MAX_STEPS];
    if (threadIdx.x < l.n) { tx[threadIdx.x] = l.x[threadIdx.x]; ty[threadIdx.x] = l.y[threadIdx.x]; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    out[i] = sh_lut(tx, ty, l.n, v[i], inverse, linear);
}

__global__ __launch_bounds__(NT) void k_lut_bwd(const float* __restrict__ v, const float* __restrict__ g_out, size_t n, Lut l, int inverse, int linear,
                                                float* __restrict__ g_in) {
    __shared__ float tx[MAX_STEPS], ty[MAX_STEPS];
    if (threadIdx.x < l.n) { tx[threadIdx.x] = l.x[threadIdx.x]; ty[threadIdx.x] = l.y[threadIdx.x]; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    g_in[i] = g_out[i] * sh_lut_grad(tx, ty, l.n, v[i], inverse, linear);
}

__global__ __launch_bounds__(NT) void k_shade_views(const float* __restrict__ rgba, const float* __restrict__ normal_fg,
                                                    const float* __restrict__ lights, unsigned n_views, unsigned pix, float ambient,
                                                    float bg, Lut l, float* __restrict__ image) {
    __shared__ float tx[MAX_STEPS], ty[MAX_STEPS];
    if (l.n > 0 && threadIdx.x < l.n) { tx[threadIdx.x] = l.x[threadIdx.x]; ty[threadIdx.x] = l.y[threadIdx.x]; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)n_views * pix) return;
    sh_shade_pixel(rgba + i * 4, normal_fg + i * 3, lights + 3 * (i / pix), ambient, bg, tx, ty, l.n, image + i * 3);
}

// forward (g_out == nullptr) or backward of sh_shade_point over N points
__global__ __launch_bounds__(NT) void k_shade_points(const float* __restrict__ albedo, const float* __restrict__ normal, const float* __restrict__ lights,
                                                     size_t N, float ambient, Lut l, float* __restrict__ out, const float* __restrict__ g_out,
                                                     float* __restrict__ g_albedo, float* __restrict__ g_normal) {
    __shared__ float tx[MAX_STEPS], ty[MAX_STEPS];
    if (l.n > 0 && threadIdx.x < l.n) { tx[threadIdx.x] = l.x[threadIdx.x]; ty[threadIdx.x] = l.y[threadIdx.x]; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    sh_shade_point(albedo + 3 * i, normal + 3 * i, lights + 3 * i, ambient, tx, ty, l.n, out ? out + 3 * i : nullptr, g_out ? g_out + 3 * i : nullptr,
                   g_albedo ? g_albedo + 3 * i : nullptr, g_normal ? g_normal + 3 * i : nullptr);
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

int mve_tonemap_lut_backward(const float* d_x, const float* d_grad_out, size_t n, const float* d_lut_x, const float* d_lut_y, int steps,
                             int inverse, int linear, float* d_grad_x, void* stream) {
    if (n == 0) return MVE_OK;
    MVE_CHECK(d_x && d_grad_out && d_grad_x && d_lut_x && d_lut_y && steps >= 2 && steps <= MAX_STEPS, MVE_ERR_ARG,
              "tonemap_lut_backward: bad arguments (2 <= steps <= %d)", MAX_STEPS);
    k_lut_bwd<<<mve_cdiv(n, NT), NT, 0, (hipStream_t)stream>>>(d_x, d_grad_out, n, Lut{d_lut_x, d_lut_y, steps}, inverse ? 1 : 0, linear ? 1 : 0,
                                                                d_grad_x);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_shade_points(const float* d_albedo, const float* d_normal, const float* d_lights, size_t N, float ambient_light, const float* d_lut_x,
                     const float* d_lut_y, int steps, float* d_out, const float* d_grad_out, float* d_grad_albedo, float* d_grad_normal, void* stream) {
    if (N == 0) return MVE_OK;
    MVE_CHECK(d_albedo && d_normal && d_lights, MVE_ERR_ARG, "shade_points: null pointer");
    MVE_CHECK((d_grad_out == nullptr) ? (d_out != nullptr) : (d_grad_albedo && d_grad_normal), MVE_ERR_ARG,
              "shade_points: forward needs out, backward needs grad_out, grad_albedo and grad_normal");
    MVE_CHECK((d_lut_x == nullptr) == (d_lut_y == nullptr) && (d_lut_x == nullptr || (steps >= 2 && steps <= MAX_STEPS)), MVE_ERR_ARG,
              "shade_points: pass both tables (2 <= steps <= %d) or neither", MAX_STEPS);
    k_shade_points<<<mve_cdiv(N, NT), NT, 0, (hipStream_t)stream>>>(d_albedo, d_normal, d_lights, N, ambient_light,
                                                                   Lut{d_lut_x, d_lut_y, d_lut_x ? steps : 0}, d_out, d_grad_out, d_grad_albedo, d_grad_normal);
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
