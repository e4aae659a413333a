// Per-pixel arithmetic of the mip-mapped texture path (texture_mip.hip), host/device so that the CPU tests can run the same source
// (oracle/devcore_host.cpp).  Replaces what the reference gets from nvdiffrast with filter_mode='linear-mipmap-linear'
// (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:196, :241, :260-264, :357-361, :442, :466-474, :496-500, :544-551, :573-577):
//   dr.rasterize(...)[1]                          rast_db  = (du/dX, du/dY, dv/dX, dv/dY) of the barycentrics per pixel       tm_rast_db
//   dr.interpolate(..., rast_db, diff_attrs='all') attribute pixel differentials (dA/dX, dA/dY per channel)                    tm_attr_da
//   dr.texture(tex, uv, uv_da, 'linear-mipmap-linear')   level from the footprint's major axis, two wrapped bilinear fetches     tm_level, tm_taps
// nvdiffrast@c5caf7b is not vendored in the reference: the algorithm is restated from its published sources (oracle/texture_mip_oracle.py
// has the derivation and the same statement in torch).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MVE_TM_FN __host__ __device__ __forceinline__
#else
#define MVE_TM_FN static inline
#endif

// p0, p1, p2: clip-space (x, y, z, w) of the pixel's triangle; (b0, b1): the barycentrics the rasteriser stored for pixel (px, py)
MVE_TM_FN void tm_rast_db(const float* p0, const float* p1, const float* p2, float b0, float b1, int px, int py, int W, int H, float* out) {
    const float xs = 2.0f / (float)W, ys = 2.0f / (float)H, xo = 1.0f / (float)W - 1.0f, yo = 1.0f / (float)H - 1.0f;
    const float fx = xs * (float)px + xo, fy = ys * (float)py + yo;
    const float p0x = p0[0] - fx * p0[3], p0y = p0[1] - fy * p0[3];
    const float p1x = p1[0] - fx * p1[3], p1y = p1[1] - fy * p1[3];
    const float p2x = p2[0] - fx * p2[3], p2y = p2[1] - fy * p2[3];
    const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
    const float iw = 1.0f / (a0 + a1 + a2);
    const float dfxdx = xs * iw, dfydy = ys * iw;
    const float da0dx = p2[1] * p1[3] - p1[1] * p2[3], da0dy = p1[0] * p2[3] - p2[0] * p1[3];
    const float da1dx = p0[1] * p2[3] - p2[1] * p0[3], da1dy = p2[0] * p0[3] - p0[0] * p2[3];
    const float da2dx = p1[1] * p0[3] - p0[1] * p1[3], da2dy = p0[0] * p1[3] - p1[0] * p0[3];
    const float datdx = da0dx + da1dx + da2dx, datdy = da0dy + da1dy + da2dy;
    out[0] = dfxdx * (b0 * datdx - da0dx);
    out[1] = dfydy * (b0 * datdy - da0dy);
    out[2] = dfxdx * (b1 * datdx - da1dx);
    out[3] = dfydy * (b1 * datdy - da1dy);
}

// one attribute channel: values a0, a1, a2 at the triangle's vertices -> (dA/dX, dA/dY)
MVE_TM_FN void tm_attr_da(float a0, float a1, float a2, const float* db, float* dx, float* dy) {
    const float dsdu = a0 - a2, dsdv = a1 - a2;
    *dx = db[0] * dsdu + db[2] * dsdv;
    *dy = db[1] * dsdu + db[3] * dsdv;
}

struct TmLevel { int l0, l1; float f; };

// da = (du/dX, du/dY, dv/dX, dv/dY) in texture-coordinate units per pixel
MVE_TM_FN TmLevel tm_level(const float* da, int tw, int th, int max_level) {
    const float dsdx = da[0] * (float)tw, dsdy = da[1] * (float)tw, dtdx = da[2] * (float)th, dtdy = da[3] * (float)th;
    const float A = dsdx * dsdx + dtdx * dtdx, B = dsdy * dsdy + dtdy * dtdy, C = dsdx * dsdy + dtdx * dtdy;
    const float major = 0.5f * (A + B) + sqrtf(0.25f * (A - B) * (A - B) + C * C);
    float lvl = 0.5f * log2f(major);                       // -inf / NaN for an empty footprint: clamped to level 0
    if (!(lvl > 0.0f)) lvl = 0.0f;
    if (lvl > (float)max_level) lvl = (float)max_level;
    TmLevel r;
    r.l0 = (int)floorf(lvl);
    r.l1 = r.l0;
    r.f = 0.0f;
    if (lvl > 0.0f) {
        r.l1 = r.l0 + 1 < max_level ? r.l0 + 1 : max_level;
        r.f = lvl - (float)r.l0;
    }
    return r;
}

MVE_TM_FN int tm_dim(int n, int level) { const int v = n >> level; return v > 0 ? v : 1; }

// wrapped bilinear taps of one level (w x h texels): indices and the two fractions
MVE_TM_FN void tm_taps(float u, float v, int w, int h, int* ix, int* iy, float* fu, float* fv) {
    u = u - floorf(u);
    v = v - floorf(v);
    u = u * (float)w - 0.5f;
    v = v * (float)h - 0.5f;
    const int iu0 = (int)floorf(u), iv0 = (int)floorf(v);
    *fu = u - (float)iu0;
    *fv = v - (float)iv0;
    int a = iu0 % w, b = (iu0 + 1) % w, c = iv0 % h, d = (iv0 + 1) % h;
    ix[0] = a < 0 ? a + w : a; ix[1] = b < 0 ? b + w : b;
    iy[0] = c < 0 ? c + h : c; iy[1] = d < 0 ? d + h : d;
}

// offset (in texels) of level l >= 1 inside the mip buffer of ONE texture: levels 1 .. l-1 come first
MVE_TM_FN long long tm_mip_offset(int H, int W, int level) {
    long long o = 0;
    for (int k = 1; k < level; ++k) o += (long long)tm_dim(H, k) * tm_dim(W, k);
    return o;
}
