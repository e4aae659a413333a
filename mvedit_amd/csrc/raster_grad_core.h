// Per-pixel bodies of the mesh geometry-gradient kernels (raster.hip: k_interpolate_bwd_rast, k_rasterize_bwd, k_antialias_bwd_pos),
// written so that the SAME source also compiles for the host: oracle/devcore_host.cpp wraps them in plain loops and the CPU tests
// compare that build with the closed forms / finite differences of oracle/raster_grad_oracle.py.  On the device MVE_RG_ADD is a float
// atomicAdd, on the host a plain add.  No fma contraction in either build (-ffp-contract=off).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MVE_RG_FN __device__ __forceinline__
#define MVE_RG_ADD(p, v) atomicAdd((p), (v))
#else
#define MVE_RG_FN static inline
#define MVE_RG_ADD(p, v) (*(p) += (v))
#endif

struct RGView { const float* rast; const float* pos; const int32_t* tri; const int32_t* opp; int V, F, H, W; };   // one view
struct RGEdge { int va, vb; bool horizontal; float s, d, ex, ey, sign; };

// d interpolate / d (u, v) of one pixel: out = u a0 + v a1 + (1 - u - v) a2
MVE_RG_FN void rg_interpolate_bwd_rast(const float* attr /* this view's [V][A] */, int A, const float* rast_px, const int32_t* tri, int F,
                                       const float* g /* [A] */, float* g_rast_px /* [4] */) {
    g_rast_px[0] = g_rast_px[1] = g_rast_px[2] = g_rast_px[3] = 0.0f;
    const int id = (int)rast_px[3] - 1;
    if (id < 0 || id >= F) return;
    const float* a0 = attr + (size_t)tri[3 * id] * A;
    const float* a1 = attr + (size_t)tri[3 * id + 1] * A;
    const float* a2 = attr + (size_t)tri[3 * id + 2] * A;
    float gu = 0.f, gv = 0.f;
    for (int a = 0; a < A; ++a) { gu += g[a] * (a0[a] - a2[a]); gv += g[a] * (a1[a] - a2[a]); }
    g_rast_px[0] = gu; g_rast_px[1] = gv;
}

// d (u, v, z/w) of pixel (px, py) / d the clip-space vertices of its triangle (continuous barycentrics), scattered into g_pos [V][4]
MVE_RG_FN void rg_rasterize_bwd(const float* pos /* this view's [V][4] */, const int32_t* tri, int F, int H, int W, const float* rast_px,
                                const float* g_px /* (gu, gv, gz, -) */, int px, int py, float* g_pos) {
    const int id = (int)rast_px[3] - 1;
    if (id < 0 || id >= F) return;
    const float gu = g_px[0], gv = g_px[1], gz = g_px[2];
    if (gu == 0.f && gv == 0.f && gz == 0.f) return;
    const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
    int vid[3];
    float x[3], y[3], z[3], iw[3], sx[3], sy[3];
    for (int k = 0; k < 3; ++k) {
        vid[k] = tri[3 * id + k];
        const float* p = pos + 4 * (size_t)vid[k];
        x[k] = p[0]; y[k] = p[1]; z[k] = p[2]; iw[k] = 1.0f / p[3];
        sx[k] = (x[k] * iw[k] * 0.5f + 0.5f) * (float)W;
        sy[k] = (y[k] * iw[k] * 0.5f + 0.5f) * (float)H;
    }
    float E[3];
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, c = (k + 2) % 3;
        E[k] = (sx[c] - sx[a]) * (cy - sy[a]) - (sy[c] - sy[a]) * (cx - sx[a]);
    }
    const float tot = E[0] + E[1] + E[2];
    if (tot == 0.f) return;
    float bq[3], q[3];
    for (int k = 0; k < 3; ++k) { bq[k] = E[k] / tot; q[k] = bq[k] * iw[k]; }
    const float S = q[0] + q[1] + q[2];
    const float u = q[0] / S, v = q[1] / S;
    const float common = gu * u + gv * v;
    const float gq[3] = {(gu - common) / S, (gv - common) / S, -common / S};
    float gb[3], g_iw[3], g_zw[3], dotb = 0.f;
    for (int k = 0; k < 3; ++k) {
        gb[k] = gq[k] * iw[k] + gz * (z[k] * iw[k]);
        g_iw[k] = gq[k] * bq[k];
        g_zw[k] = gz * bq[k];
        dotb += gb[k] * bq[k];
    }
    float gsx[3] = {0.f, 0.f, 0.f}, gsy[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, c = (k + 2) % 3;
        const float gE = (gb[k] - dotb) / tot;
        gsx[a] += gE * (sy[c] - cy);
        gsy[a] += gE * (cx - sx[c]);
        gsx[c] += gE * (cy - sy[a]);
        gsy[c] += -gE * (cx - sx[a]);
    }
    for (int k = 0; k < 3; ++k) {
        float* dst = g_pos + 4 * (size_t)vid[k];
        const float hx = gsx[k] * (0.5f * (float)W), hy = gsy[k] * (0.5f * (float)H);
        MVE_RG_ADD(dst + 0, hx * iw[k]);
        MVE_RG_ADD(dst + 1, hy * iw[k]);
        MVE_RG_ADD(dst + 2, g_zw[k] * iw[k]);
        MVE_RG_ADD(dst + 3, -(hx * x[k] + hy * y[k] + g_zw[k] * z[k] + g_iw[k]) * iw[k] * iw[k]);
    }
}

// the antialias pair rule (raster.hip: aa_pair; oracle/raster_oracle.c: aa_pair) with the crossed edge reported
MVE_RG_FN bool rg_aa_pair(const RGView& a, int px, int py, int qx, int qy, bool& dst_is_p, float& wgt, RGEdge& ed) {
    const float* rp = a.rast + 4 * ((size_t)py * a.W + px);
    const float* rq = a.rast + 4 * ((size_t)qy * a.W + qx);
    const int ip = (int)rp[3] - 1, iq = (int)rq[3] - 1;
    if (ip == iq) return false;
    bool use_p;
    if (ip < 0) use_p = false;
    else if (iq < 0) use_p = true;
    else if (rp[2] != rq[2]) use_p = rp[2] < rq[2];
    else use_p = (py * a.W + px) < (qy * a.W + qx);
    const int t = use_p ? ip : iq;
    if (t < 0 || t >= a.F) return false;
    const int ox = use_p ? px : qx, oy = use_p ? py : qy, nx = use_p ? qx : px, ny = use_p ? qy : py;
    float sx[3], sy[3];
    int vi[3];
    for (int k = 0; k < 3; ++k) {
        vi[k] = a.tri[3 * t + k];
        if (vi[k] < 0 || vi[k] >= a.V) return false;
        const float* v = a.pos + 4 * (size_t)vi[k];
        if (v[3] <= 1e-6f) return false;
        sx[k] = (v[0] / v[3] * 0.5f + 0.5f) * (float)a.W;
        sy[k] = (v[1] / v[3] * 0.5f + 0.5f) * (float)a.H;
    }
    const float cx = (float)ox + 0.5f, cy = (float)oy + 0.5f;
    const float dx = (float)(nx - ox), dy = (float)(ny - oy);
    float best = 2.0f;
    for (int e = 0; e < 3; ++e) {
        const int ia = e, ib = (e + 1) % 3, ic = (e + 2) % 3;
        const float ex = sx[ib] - sx[ia], ey = sy[ib] - sy[ia];
        const int o = a.opp[3 * t + e];
        if (o >= 0) {
            if (o >= a.V) continue;
            const float* v = a.pos + 4 * (size_t)o;
            if (v[3] <= 1e-6f) continue;
            const float oxs = (v[0] / v[3] * 0.5f + 0.5f) * (float)a.W, oys = (v[1] / v[3] * 0.5f + 0.5f) * (float)a.H;
            const float sc = ex * (sy[ic] - sy[ia]) - ey * (sx[ic] - sx[ia]);
            const float so = ex * (oys - sy[ia]) - ey * (oxs - sx[ia]);
            if (!(sc * so >= 0.0f)) continue;
        }
        float s, tt;
        if (dy == 0.0f) {
            if (ey == 0.0f) continue;
            s = (cy - sy[ia]) / ey;
            tt = ((sx[ia] + s * ex) - cx) * dx;
        } else {
            if (ex == 0.0f) continue;
            s = (cx - sx[ia]) / ex;
            tt = ((sy[ia] + s * ey) - cy) * dy;
        }
        if (s >= 0.0f && s <= 1.0f && tt >= 0.0f && tt <= 1.0f && tt < best) {
            best = tt;
            ed.va = vi[ia]; ed.vb = vi[ib]; ed.horizontal = dy == 0.0f; ed.s = s; ed.d = dy == 0.0f ? dx : dy; ed.ex = ex; ed.ey = ey;
        }
    }
    if (best > 1.0f) return false;
    if (best > 0.5f) { dst_is_p = !use_p; wgt = best - 0.5f; ed.sign = 1.0f; }
    else if (best < 0.5f) { dst_is_p = use_p; wgt = 0.5f - best; ed.sign = -1.0f; }
    else return false;
    return true;
}

// silhouette gradient of pixel (x, y) as the RECEIVING pixel of its blends: d out / d the crossed edges' vertices -> g_pos [V][4]
MVE_RG_FN void rg_antialias_bwd_pos(const RGView& a, const float* col /* this view's [H*W][C] */, const float* g /* d out[(x,y)] [C] */, int C,
                                    int x, int y, float* g_pos) {
    const float* self = col + ((size_t)y * a.W + x) * C;
    const int ddx[4] = {-1, 1, 0, 0}, ddy[4] = {0, 0, -1, 1};
    for (int k = 0; k < 4; ++k) {
        const int qx = x + ddx[k], qy = y + ddy[k];
        if (qx < 0 || qx >= a.W || qy < 0 || qy >= a.H) continue;
        bool dst_is_p;
        float w;
        RGEdge ed;
        if (!rg_aa_pair(a, x, y, qx, qy, dst_is_p, w, ed) || !dst_is_p) continue;     // each blend is visited once, from its receiving pixel
        const float* nb = col + ((size_t)qy * a.W + qx) * C;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot += g[c] * (nb[c] - self[c]);
        const float g_tt = ed.sign * dot;
        if (g_tt == 0.f) continue;
        float gsx[2], gsy[2];
        if (ed.horizontal) {
            gsx[0] = (1.0f - ed.s) * ed.d; gsx[1] = ed.s * ed.d;
            gsy[0] = ed.ex * (ed.s - 1.0f) / ed.ey * ed.d; gsy[1] = -ed.ex * ed.s / ed.ey * ed.d;
        } else {
            gsy[0] = (1.0f - ed.s) * ed.d; gsy[1] = ed.s * ed.d;
            gsx[0] = ed.ey * (ed.s - 1.0f) / ed.ex * ed.d; gsx[1] = -ed.ey * ed.s / ed.ex * ed.d;
        }
        const int vid[2] = {ed.va, ed.vb};
        for (int j = 0; j < 2; ++j) {
            const float* p = a.pos + 4 * (size_t)vid[j];
            const float iw = 1.0f / p[3];
            const float hx = g_tt * gsx[j] * (0.5f * (float)a.W), hy = g_tt * gsy[j] * (0.5f * (float)a.H);
            MVE_RG_ADD(g_pos + 4 * (size_t)vid[j] + 0, hx * iw);
            MVE_RG_ADD(g_pos + 4 * (size_t)vid[j] + 1, hy * iw);
            MVE_RG_ADD(g_pos + 4 * (size_t)vid[j] + 3, -(hx * p[0] + hy * p[1]) * iw * iw);
        }
    }
}
