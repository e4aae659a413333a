// Separable Gaussian blur with reflect padding and its adjoint (blur.hip), host/device like the other *_core.h headers.
// Reference: torchvision.transforms.functional.gaussian_blur as the pipelines call it -- `highpass` (lib/pipelines/utils.py:187-188:
// offset + x - gaussian_blur(x, 6 round(std) + 1, std), applied to predicted and target normal patches every optimisation iteration,
// lib/pipelines/mvedit_3d_pipeline.py:623-624, :815-816) and the target-mask blur (:473, :671).  torchvision (absent here) pads with
// mode='reflect' by ksize // 2 and correlates every channel with the outer product of the normalised 1-D kernel; the two 1-D passes below
// are that product evaluated separably (differences: fp32 summation order only).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MVE_BL_FN __device__ __forceinline__
#else
#define MVE_BL_FN static inline
#endif

// torchvision's _get_gaussian_kernel1d in float32: x = linspace(-(k-1)/2, (k-1)/2, k), pdf = exp(-0.5 (x / sigma)^2), pdf / sum(pdf)
static inline void bl_kernel1d(int ksize, float sigma, float* w) {
    const float half = (float)(ksize - 1) * 0.5f;
    float sum = 0.f;
    for (int t = 0; t < ksize; ++t) {
        const float x = ksize > 1 ? -half + (2.0f * half) * (float)t / (float)(ksize - 1) : 0.f;
        const float q = x / sigma;
        w[t] = expf(-0.5f * (q * q));
        sum += w[t];
    }
    for (int t = 0; t < ksize; ++t) w[t] /= sum;
}

// source index of padded position j (mode='reflect': the edge sample is not repeated)
MVE_BL_FN int bl_reflect(int j, int n) { return j < 0 ? -j : (j >= n ? 2 * (n - 1) - j : j); }

// out[i] = sum_t w[t] x[reflect(i + t - r)] along a line of n samples with the given stride
MVE_BL_FN float bl_fwd1d(const float* x, int n, int stride, const float* w, int r, int i) {
    float acc = 0.f;
    for (int t = 0; t <= 2 * r; ++t) acc += w[t] * x[(size_t)bl_reflect(i + t - r, n) * stride];
    return acc;
}

// adjoint of bl_fwd1d as a gather: g_x[i] = sum over (o, t) with reflect(o + t - r) == i of w[t] g[o].  For every tap the pre-images of i
// are o + t - r = i (direct), = -i (reflected at the left edge, only if that is < 0) and = 2(n-1) - i (right edge, only if >= n).
MVE_BL_FN float bl_adj1d(const float* g, int n, int stride, const float* w, int r, int i) {
    float acc = 0.f;
    for (int t = 0; t <= 2 * r; ++t) {
        const int d = t - r;
        int o = i - d;
        if (o >= 0 && o < n) acc += w[t] * g[(size_t)o * stride];
        o = -i - d;                                   // padded position j = -i < 0
        if (i > 0 && o >= 0 && o < n) acc += w[t] * g[(size_t)o * stride];
        o = 2 * (n - 1) - i - d;                      // padded position j = 2(n-1) - i >= n
        if (i < n - 1 && o >= 0 && o < n) acc += w[t] * g[(size_t)o * stride];
    }
    return acc;
}
