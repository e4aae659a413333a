// TEST INFRASTRUCTURE ONLY.  Host build of the product's host/device "core" headers (mvedit_amd/csrc/raster_grad_core.h,
// shading_core.h): the per-pixel bodies of kernels whose first GPU run is still pending are wrapped in plain loops here, so that the
// CPU tests can run that very arithmetic against the Python closed forms, finite differences of the C oracle, and outputs of the
// reference's Tonemapping class.  Built by oracle.build_devcore() with g++ -ffp-contract=off into oracle/libdevcore.so.
#include <stddef.h>
#include <stdint.h>

#include "../mvedit_amd/csrc/raster_grad_core.h"
#include "../mvedit_amd/csrc/shading_core.h"

extern "C" {

void dc_interpolate_backward_rast(const float* attr, int Battr, int Vattr, int A, const float* rast, int B, int H, int W, const int32_t* tri,
                                  int F, const float* g_out, float* g_rast) {
    const size_t npix = (size_t)H * W;
    for (size_t i = 0; i < (size_t)B * npix; ++i) {
        const int b = (int)(i / npix);
        rg_interpolate_bwd_rast(attr + (Battr > 1 ? (size_t)b * Vattr * A : 0), A, rast + 4 * i, tri, F, g_out + i * A, g_rast + 4 * i);
    }
}

void dc_rasterize_backward(const float* pos, int B, int V, const int32_t* tri, int F, int H, int W, const float* rast, const float* g_rast,
                           float* g_pos) {
    const size_t npix = (size_t)H * W;
    for (size_t i = 0; i < (size_t)B * npix; ++i) {
        const int b = (int)(i / npix);
        rg_rasterize_bwd(pos + (size_t)b * V * 4, tri, F, H, W, rast + 4 * i, g_rast + 4 * i, (int)(i % W), (int)((i % npix) / W), g_pos + (size_t)b * V * 4);
    }
}

void dc_antialias_backward_pos(const float* color, const float* g_out, int B, int H, int W, int C, const float* rast, const float* pos, int V,
                               const int32_t* tri, int F, const int32_t* opp, float* g_pos) {
    const size_t npix = (size_t)H * W;
    for (size_t i = 0; i < (size_t)B * npix; ++i) {
        const int b = (int)(i / npix), y = (int)((i % npix) / W), x = (int)(i % W);
        const RGView a{rast + (size_t)b * npix * 4, pos + (size_t)b * V * 4, tri, opp, V, F, H, W};
        rg_antialias_bwd_pos(a, color + (size_t)b * npix * C, g_out + i * C, C, x, y, g_pos + (size_t)b * V * 4);
    }
}

void dc_tonemap_lut(const float* x, size_t n, const float* lut_x, const float* lut_y, int steps, int inverse, int linear, float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = sh_lut(lut_x, lut_y, steps, x[i], inverse, linear);
}

void dc_shade_views(const float* rgba, const float* normal_fg, const float* lights, unsigned n_views, unsigned pix, float ambient, float bg,
                    const float* lut_x, const float* lut_y, int steps, float* image) {
    for (size_t i = 0; i < (size_t)n_views * pix; ++i)
        sh_shade_pixel(rgba + i * 4, normal_fg + i * 3, lights + 3 * (i / pix), ambient, bg, lut_x, lut_y, lut_x ? steps : 0, image + i * 3);
}

}  // extern "C"
