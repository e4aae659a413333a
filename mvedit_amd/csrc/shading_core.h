// Per-element arithmetic of the tone-mapping table and the render-step shading (shading.hip), written so that the same source also
// compiles for the host (oracle/devcore_host.cpp): the CPU tests compare that build with outputs of the reference's Tonemapping class.
// No fma contraction in either build.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MVE_SH_FN __device__ __forceinline__
#else
#define MVE_SH_FN static inline
#endif

// torch.bucketize(v, table, right=True).clamp(1, n - 1): number of table entries <= v
MVE_SH_FN int sh_bucket(const float* t, int n, float v) {
    int i = 0;
    for (int k = 0; k < n; ++k) i += (t[k] <= v) ? 1 : 0;
    return i < 1 ? 1 : (i > n - 1 ? n - 1 : i);
}

// piecewise-linear map from table a to table b (lut: a = lut_x, b = lut_y; inverse_lut: a = lut_y, b = lut_x)
MVE_SH_FN float sh_interp(const float* a, const float* b, int n, float v) {
    const int i = sh_bucket(a, n, v);
    const float t = (v - a[i - 1]) / (a[i] - a[i - 1]);
    return b[i - 1] + (b[i] - b[i - 1]) * t;
}

// Tonemapping.lut (inverse = 0; linear: input_mode='linear') / Tonemapping.inverse_lut (inverse = 1; linear: output_mode='linear')
MVE_SH_FN float sh_lut(const float* tx, const float* ty, int n, float x, int inverse, int linear) {
    if (!inverse) {
        if (linear) x = log2f(fmaxf(x, 1e-6f));
        return sh_interp(tx, ty, n, x);
    }
    const float r = sh_interp(ty, tx, n, x);
    return linear ? exp2f(r) : r;
}

// d sh_lut / d x (what torch autograd gives the reference's lut / inverse_lut: the slope of the selected segment, times the derivative of
// the log2 / exp2 wrapped around it; clamp(min=1e-6) passes the gradient for x >= 1e-6)
MVE_SH_FN float sh_lut_grad(const float* tx, const float* ty, int n, float x, int inverse, int linear) {
    if (!inverse) {
        float d = 1.0f;
        if (linear) {
            d = x >= 1e-6f ? 1.0f / (fmaxf(x, 1e-6f) * 0.69314718055994531f) : 0.0f;
            x = log2f(fmaxf(x, 1e-6f));
        }
        const int i = sh_bucket(tx, n, x);
        return d * ((ty[i] - ty[i - 1]) / (tx[i] - tx[i - 1]));
    }
    const int i = sh_bucket(ty, n, x);
    const float slope = (tx[i] - tx[i - 1]) / (ty[i] - ty[i - 1]);
    return linear ? exp2f(sh_interp(ty, tx, n, x)) * 0.69314718055994531f * slope : slope;
}

// one pixel of lib/pipelines/mvedit_3d_pipeline.py:1372-1384 (n = 0: `self.tonemapping is None`)
MVE_SH_FN void sh_shade_pixel(const float* rgba, const float* normal_fg, const float* light, float ambient, float bg, const float* tx,
                              const float* ty, int n, float* out) {
    const float n0 = normal_fg[0] * 2.0f - 1.0f, n1 = -normal_fg[1] * 2.0f + 1.0f, n2 = -normal_fg[2] * 2.0f + 1.0f;
    const float dot = (light[0] * n0 + light[1] * n1) + light[2] * n2;
    const float shading = fmaxf(dot, 0.0f) * (1.0f - ambient) + ambient;
    const float a = rgba[3], back = bg * (1.0f - a);
    if (n > 0) {
        const float ls = log2f(fmaxf(shading, 1e-6f)), den = fmaxf(a, 1e-6f);
        for (int k = 0; k < 3; ++k) out[k] = sh_interp(tx, ty, n, sh_interp(ty, tx, n, rgba[k] / den) + ls) * a + back;
    } else {
        for (int k = 0; k < 3; ++k) out[k] = rgba[k] * shading + back;
    }
}

// ---- per-point Lambertian shading of the mesh path: the `shading_fun`s handed to MeshRenderer.forward (lib/pipelines/mvedit_3d_pipeline.py:
// 410-440 make_shading_fun / make_nerf_shading_fun):  shading = clamp(light . normal, 0) (1 - ambient) + ambient;
//   tables given : out = lut(inverse_lut(albedo) + log2(clamp(shading, 1e-6)))        tables absent (n = 0): out = albedo * shading
// Backward (g_out != nullptr): g_albedo[3], g_normal[3] of sum_k g_out[k] out[k]; the light is a constant.
MVE_SH_FN void sh_shade_point(const float* albedo, const float* normal, const float* light, float ambient, const float* tx, const float* ty, int n,
                              float* out, const float* g_out, float* g_albedo, float* g_normal) {
    const float dot = (light[0] * normal[0] + light[1] * normal[1]) + light[2] * normal[2];
    const float shading = fmaxf(dot, 0.0f) * (1.0f - ambient) + ambient;
    float g_shading = 0.f;
    if (n > 0) {
        const float sc = fmaxf(shading, 1e-6f), ls = log2f(sc);
        float g_ls = 0.f;
        for (int k = 0; k < 3; ++k) {
            const int ii = sh_bucket(ty, n, albedo[k]);
            const float s_inv = (tx[ii] - tx[ii - 1]) / (ty[ii] - ty[ii - 1]);
            const float u = tx[ii - 1] + (tx[ii] - tx[ii - 1]) * ((albedo[k] - ty[ii - 1]) / (ty[ii] - ty[ii - 1])) + ls;
            const int io = sh_bucket(tx, n, u);
            const float s_lut = (ty[io] - ty[io - 1]) / (tx[io] - tx[io - 1]);
            if (out) out[k] = ty[io - 1] + (ty[io] - ty[io - 1]) * ((u - tx[io - 1]) / (tx[io] - tx[io - 1]));
            if (g_out) { g_albedo[k] = g_out[k] * s_lut * s_inv; g_ls += g_out[k] * s_lut; }
        }
        if (g_out && shading >= 1e-6f) g_shading = g_ls / (sc * 0.69314718055994531f);
    } else {
        for (int k = 0; k < 3; ++k) {
            if (out) out[k] = albedo[k] * shading;
            if (g_out) { g_albedo[k] = g_out[k] * shading; g_shading += g_out[k] * albedo[k]; }
        }
    }
    if (g_out) {
        const float gd = dot >= 0.0f ? g_shading * (1.0f - ambient) : 0.0f;
        for (int k = 0; k < 3; ++k) g_normal[k] = gd * light[k];
    }
}
