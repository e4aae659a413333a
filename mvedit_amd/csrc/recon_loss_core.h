// Per-pixel / per-sample arithmetic of the image-space loss of one NeRF optimisation iteration and its gradients (recon_loss.hip),
// written so that the same source also compiles for the host (oracle/devcore_host.cpp): the CPU tests run that build against the
// reference's own statements (tests/golden/recon_loss_ref.npz).  Reference: lib/pipelines/mvedit_3d_pipeline.py:542-603 (nerf_optim),
// lib/core/utils/geometry_utils.py:119-148 (depth_to_normal), lib/models/losses/tv_loss.py:8-40 (power 1.5 over dims (-2, -1)),
// lib/models/losses/pixelwise_loss.py:10-21 (l1_loss_mod), lib/models/decoders/tonemapping.py:33-53.
//
// Pixels are patch-major: p = (n * ps + y) * ps + x.  Every function is a pure gather over read-only arrays, so the device build needs
// no atomics and its results do not depend on the launch geometry.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MVE_RL_FN __device__ __forceinline__
#else
#define MVE_RL_FN static inline
#endif

struct RlParams {
    int P, ps;                   // patches, patch side
    int shaded;                  // `not is_init or init_shaded`
    int lut_n;                   // tone-mapping knots (0: `self.tonemapping is None`)
    float ambient, bg, nbg[3];   // ambient_light, nerf.bg_color, normal_bg
    float c_rgb, c_alpha, c_tv, c_depth, c_ent, log_bgw;   // loss coefficients incl. the 1 / numel of the means; log(bg_width)
};

// coefficients from the reference's hyper-parameters: pixel_loss = L1LossMod(loss_weight) over P*ps*ps*C elements (mean), * 4.5 for the
// colours (:574-576), * (5 if is_init else 1) for alpha (:578-580), * depth_weight (:589-591); TVLoss mean over P*3*ps*ps, *
// normal_reg_weight * 10 (:581-584); entropy * entropy_weight / (P*ps*ps) (:598-602).
static inline RlParams rl_make_params(int P, int ps, int shaded, int is_init, int lut_n, float ambient, float bg, const float* normal_bg,
                                      float pixel_loss_weight, float normal_reg_weight, float depth_weight, float entropy_weight, float bg_width) {
    RlParams q;
    const double N = (double)P * ps * ps;
    q.P = P; q.ps = ps; q.shaded = shaded; q.lut_n = lut_n; q.ambient = ambient; q.bg = bg;
    for (int k = 0; k < 3; ++k) q.nbg[k] = normal_bg[k];
    q.c_rgb = (float)(pixel_loss_weight * 4.5 / (3.0 * N));
    q.c_alpha = (float)(pixel_loss_weight * (is_init ? 5.0 : 1.0) / N);
    q.c_tv = (float)(normal_reg_weight * 10.0 / (3.0 * N));
    q.c_depth = (float)(pixel_loss_weight * depth_weight / N);
    q.c_ent = (float)(entropy_weight / N);
    q.log_bgw = (float)log((double)bg_width);
    return q;
}

struct RlV3 { float x, y, z; };
MVE_RL_FN RlV3 rl_v(float x, float y, float z) { RlV3 r; r.x = x; r.y = y; r.z = z; return r; }
MVE_RL_FN RlV3 rl_ld(const float* a, int p) { return rl_v(a[3 * p], a[3 * p + 1], a[3 * p + 2]); }
MVE_RL_FN void rl_st(float* a, int p, RlV3 v) { a[3 * p] = v.x; a[3 * p + 1] = v.y; a[3 * p + 2] = v.z; }
MVE_RL_FN RlV3 rl_add(RlV3 a, RlV3 b) { return rl_v(a.x + b.x, a.y + b.y, a.z + b.z); }
MVE_RL_FN RlV3 rl_sub(RlV3 a, RlV3 b) { return rl_v(a.x - b.x, a.y - b.y, a.z - b.z); }
MVE_RL_FN RlV3 rl_mul(RlV3 a, float s) { return rl_v(a.x * s, a.y * s, a.z * s); }
MVE_RL_FN float rl_dot(RlV3 a, RlV3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MVE_RL_FN RlV3 rl_cross(RlV3 a, RlV3 b) { return rl_v(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MVE_RL_FN float rl_sign(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// F.normalize: v / max(|v|, 1e-12); *len receives |v|
MVE_RL_FN RlV3 rl_nrm(RlV3 v, float* len) {
    const float l = sqrtf(rl_dot(v, v));
    *len = l;
    return rl_mul(v, 1.0f / fmaxf(l, 1e-12f));
}
// gradient of rl_nrm w.r.t. v, given its output n and |v|
MVE_RL_FN RlV3 rl_nrm_bwd(RlV3 n, float len, RlV3 g) {
    const float inv = 1.0f / fmaxf(len, 1e-12f);
    if (len >= 1e-12f) return rl_mul(rl_sub(g, rl_mul(n, rl_dot(n, g))), inv);
    return rl_mul(g, inv);
}

// ---- pass 1: camera-space point of a pixel (the argument of depth_to_normal's differences) ----------------------------------------
// out_depth = depth * |dir| (:547-548), out_depth_fg = out_depth / clamp(alpha, 1e-6) (:550-551), xyz = dir / clamp(depth_fg, 1e-6)
MVE_RL_FN RlV3 rl_xyz(const float* depth, const float* alpha, const float* dir, int p) {
    const RlV3 d = rl_ld(dir, p);
    const float od = depth[p] * sqrtf(rl_dot(d, d));
    const float dfg = od / fmaxf(alpha[p], 1e-6f);
    return rl_mul(d, 1.0f / fmaxf(dfg, 1e-6f));
}

// the four difference vectors of depth_to_normal at (y, x) of patch n (replicate padding = clamped pair index)
struct RlStencil { RlV3 r, u, l, d; };
MVE_RL_FN RlStencil rl_stencil(const float* xyz, int ps, int n, int y, int x) {
    const int base = n * ps * ps;
    const int xr = x < ps - 2 ? x : ps - 2, xl = x > 1 ? x : 1, yu = y > 1 ? y : 1, yd = y < ps - 2 ? y : ps - 2;
    RlStencil s;
    s.r = rl_sub(rl_ld(xyz, base + y * ps + xr + 1), rl_ld(xyz, base + y * ps + xr));
    s.l = rl_sub(rl_ld(xyz, base + y * ps + xl - 1), rl_ld(xyz, base + y * ps + xl));
    s.u = rl_sub(rl_ld(xyz, base + (yu - 1) * ps + x), rl_ld(xyz, base + yu * ps + x));
    s.d = rl_sub(rl_ld(xyz, base + (yd + 1) * ps + x), rl_ld(xyz, base + yd * ps + x));
    return s;
}

// depth_to_normal at one pixel (opengl format, mapped to [0, 1])
MVE_RL_FN RlV3 rl_normal_fg(const RlStencil& s) {
    float t;
    const RlV3 sum = rl_add(rl_add(rl_nrm(rl_cross(s.r, s.u), &t), rl_nrm(rl_cross(s.u, s.l), &t)),
                            rl_add(rl_nrm(rl_cross(s.l, s.d), &t), rl_nrm(rl_cross(s.d, s.r), &t)));
    const RlV3 n = rl_nrm(sum, &t);
    return rl_v(n.x * 0.5f + 0.5f, -n.y * 0.5f + 0.5f, -n.z * 0.5f + 0.5f);
}

// gradient of rl_normal_fg w.r.t. the four difference vectors, given d loss / d normal_fg
MVE_RL_FN void rl_normal_fg_bwd(const RlStencil& s, RlV3 g_nfg, RlStencil* gs) {
    const RlV3 c0 = rl_cross(s.r, s.u), c1 = rl_cross(s.u, s.l), c2 = rl_cross(s.l, s.d), c3 = rl_cross(s.d, s.r);
    float l0, l1, l2, l3, ls;
    const RlV3 n0 = rl_nrm(c0, &l0), n1 = rl_nrm(c1, &l1), n2 = rl_nrm(c2, &l2), n3 = rl_nrm(c3, &l3);
    const RlV3 sum = rl_add(rl_add(n0, n1), rl_add(n2, n3));
    const RlV3 n = rl_nrm(sum, &ls);
    const RlV3 gn = rl_v(g_nfg.x * 0.5f, -g_nfg.y * 0.5f, -g_nfg.z * 0.5f);
    const RlV3 gsum = rl_nrm_bwd(n, ls, gn);
    const RlV3 g0 = rl_nrm_bwd(n0, l0, gsum), g1 = rl_nrm_bwd(n1, l1, gsum), g2 = rl_nrm_bwd(n2, l2, gsum), g3 = rl_nrm_bwd(n3, l3, gsum);
    // c = a x b:  g_a = b x g_c,  g_b = g_c x a
    gs->r = rl_add(rl_cross(s.u, g0), rl_cross(g3, s.d));
    gs->u = rl_add(rl_cross(g0, s.r), rl_cross(s.l, g1));
    gs->l = rl_add(rl_cross(g1, s.u), rl_cross(s.d, g2));
    gs->d = rl_add(rl_cross(g2, s.l), rl_cross(s.r, g3));
}

// -F.max_pool2d(-alpha, 3, stride 1, padding 1): minimum of alpha over the in-patch 3 x 3 neighbourhood (:554-556)
MVE_RL_FN float rl_min_pool(const float* alpha, int ps, int n, int y, int x) {
    float m = INFINITY;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < ps && xx >= 0 && xx < ps) m = fminf(m, alpha[(n * ps + yy) * ps + xx]);
        }
    return m;
}

// ---- tone mapping: piecewise-linear table a -> b and its slope (torch.bucketize(v, a, right=True).clamp(1, n - 1)) ----------------
MVE_RL_FN float rl_interp(const float* a, const float* b, int n, float v, float* slope) {
    int i = 0;
    for (int k = 0; k < n; ++k) i += (a[k] <= v) ? 1 : 0;
    i = i < 1 ? 1 : (i > n - 1 ? n - 1 : i);
    const float da = a[i] - a[i - 1], db = b[i] - b[i - 1];
    *slope = db / da;
    return b[i - 1] + db * ((v - a[i - 1]) / da);
}

// ---- shaded colour of a pixel (:558-572) and, when g_out != nullptr, the gradients of sum_k g_out[k] * out[k] ------------------------
// g_img[3], *g_alpha (accumulated), *g_shading out.  Returns out_rgbs in out[3].
MVE_RL_FN void rl_colour(const RlParams& q, const float* lut_x, const float* lut_y, RlV3 img, float a, float shading, float* out,
                         const float* g_out, float* g_img, float* g_alpha, float* g_shading) {
    const float back = q.bg * (1.0f - a);
    const float c[3] = {img.x, img.y, img.z};
    float ga = 0.f, gs = 0.f;
    if (!q.shaded) {
        for (int k = 0; k < 3; ++k) {
            out[k] = c[k] + back;
            if (g_out) { g_img[k] = g_out[k]; ga -= g_out[k] * q.bg; }
        }
    } else if (q.lut_n == 0) {
        for (int k = 0; k < 3; ++k) {
            out[k] = c[k] * shading + back;
            if (g_out) { g_img[k] = g_out[k] * shading; gs += g_out[k] * c[k]; ga -= g_out[k] * q.bg; }
        }
    } else {
        const float ac = fmaxf(a, 1e-6f), sc = fmaxf(shading, 1e-6f), ls = log2f(sc);
        float g_ls = 0.f;
        for (int k = 0; k < 3; ++k) {
            float s_inv, s_lut;
            const float u = rl_interp(lut_y, lut_x, q.lut_n, c[k] / ac, &s_inv) + ls;
            const float t = rl_interp(lut_x, lut_y, q.lut_n, u, &s_lut);
            out[k] = t * a + back;
            if (g_out) {
                const float gu = g_out[k] * a * s_lut;
                ga += g_out[k] * (t - q.bg);
                g_img[k] = gu * s_inv / ac;
                if (a >= 1e-6f) ga -= gu * s_inv * c[k] / (ac * ac);
                g_ls += gu;
            }
        }
        if (g_out && shading >= 1e-6f) gs = g_ls / (sc * 0.69314718055994531f);
    }
    if (g_out) { *g_alpha += ga; *g_shading = gs; }
}

// shading scalar of a pixel from its foreground normal (:559-563); *dot receives light . n_opencv
MVE_RL_FN float rl_shading(const RlParams& q, RlV3 nfg, RlV3 light, float* dot) {
    const float d = light.x * (nfg.x * 2.0f - 1.0f) + light.y * (-nfg.y * 2.0f + 1.0f) + light.z * (-nfg.z * 2.0f + 1.0f);
    *dot = d;
    return fmaxf(d, 0.0f) * (1.0f - q.ambient) + q.ambient;
}

// ---- pass 2 (forward): everything of a pixel that needs no neighbouring normals ---------------------------------------------------
// writes nfg, wfg, out_rgbs, out_normals; returns the pixel's loss terms in part[0..3] = rgb, alpha, depth, entropy (background bin)
MVE_RL_FN void rl_pixel_fwd(const RlParams& q, const float* lut_x, const float* lut_y, const float* xyz, const float* image,
                            const float* alpha, const float* depth, const float* dir, const float* tgt_rgb, const float* tgt_m,
                            const float* tgt_depth, const float* patch_w, const float* patch_light, int p, float* nfg_out, float* wfg_out,
                            float* out_rgbs, float* out_normals, float* part) {
    const int x = p % q.ps, y = (p / q.ps) % q.ps, n = p / (q.ps * q.ps);
    const RlV3 nfg = rl_normal_fg(rl_stencil(xyz, q.ps, n, y, x));
    rl_st(nfg_out, p, nfg);
    wfg_out[p] = rl_min_pool(alpha, q.ps, n, y, x);
    const float a = alpha[p], w = patch_w[n];
    rl_st(out_normals, p, rl_v(nfg.x * a + q.nbg[0] * (1.0f - a), nfg.y * a + q.nbg[1] * (1.0f - a), nfg.z * a + q.nbg[2] * (1.0f - a)));
    float dot, o[3];
    const float shading = q.shaded ? rl_shading(q, nfg, rl_ld(patch_light, n), &dot) : 1.0f;
    rl_colour(q, lut_x, lut_y, rl_ld(image, p), a, shading, o, nullptr, nullptr, nullptr, nullptr);
    float lrgb = 0.f;
    for (int k = 0; k < 3; ++k) { out_rgbs[3 * p + k] = o[k]; lrgb += fabsf(o[k] - tgt_rgb[3 * p + k]); }
    part[0] = q.c_rgb * w * lrgb;
    part[1] = q.c_alpha * w * fabsf(a - tgt_m[p]);
    part[2] = 0.f;
    if (tgt_depth) {
        const RlV3 d = rl_ld(dir, p);
        part[2] = q.c_depth * w * fabsf(depth[p] * sqrtf(rl_dot(d, d)) - tgt_depth[p]);
    }
    const float b = 1.0f - a;
    part[3] = -q.c_ent * b * (logf(fmaxf(b, 1e-6f)) - q.log_bgw);
}

// ---- TV term of a pixel (tv_loss.py; weights = min of the pooled alpha of the two pixels of a difference) --------------------------
// per channel: dh = (N(y+1,x) - N(y,x) - (T(y+1,x) - T(y,x))) * min(W(y,x), W(y+1,x)) (0 in the last row), dw likewise; term = (dh^2 + dw^2)^0.75
// Returns the sum over channels; gh[3] / gw[3] (optional) receive d term / d dh and d term / d dw.
MVE_RL_FN float rl_tv_term(const float* nfg, const float* wfg, const float* tgt_n, int ps, int n, int y, int x, float* wh_out, float* ww_out,
                           float* gh, float* gw) {
    const int p = (n * ps + y) * ps + x;
    const bool hy = y < ps - 1, hx = x < ps - 1;
    const float wh = hy ? fminf(wfg[p], wfg[p + ps]) : 0.f, ww = hx ? fminf(wfg[p], wfg[p + 1]) : 0.f;
    if (wh_out) { *wh_out = wh; *ww_out = ww; }
    float sum = 0.f;
    for (int k = 0; k < 3; ++k) {
        float dh = 0.f, dw = 0.f;
        if (hy) dh = nfg[3 * (p + ps) + k] - nfg[3 * p + k];
        if (hx) dw = nfg[3 * (p + 1) + k] - nfg[3 * p + k];
        if (tgt_n) {
            if (hy) dh -= tgt_n[3 * (p + ps) + k] - tgt_n[3 * p + k];
            if (hx) dw -= tgt_n[3 * (p + 1) + k] - tgt_n[3 * p + k];
        }
        dh *= wh; dw *= ww;
        const float r = sqrtf(dh * dh + dw * dw);
        sum += r * sqrtf(r);
        if (gh) {   // d r^1.5 / d dh = 1.5 r^0.5 * dh / r  (0 at r = 0, as torch's norm backward)
            const float f = r > 0.f ? 1.5f / sqrtf(r) : 0.f;
            gh[k] = f * dh; gw[k] = f * dw;
        }
    }
    return sum;
}

// ---- pass 1 (backward): d loss / d normal_fg of a pixel, then through depth_to_normal to its four difference vectors ---------------
// g_rgb_ext / g_nrm_ext: optional gradients arriving at out_rgbs / out_normals from the patch losses; gl = d total / d loss.
// Writes g_image[3p..], g_alpha_part[p] (everything except the path through the normals), gdir[12p..] (g_right, g_up, g_left, g_down).
MVE_RL_FN void rl_pixel_bwd(const RlParams& q, const float* lut_x, const float* lut_y, const float* xyz, const float* nfg_a, const float* wfg,
                            const float* image, const float* alpha, const float* tgt_rgb, const float* tgt_m, const float* tgt_n,
                            const float* patch_w, const float* patch_light, const float* g_rgb_ext, const float* g_nrm_ext, float gl, int p,
                            float* g_image, float* g_alpha_part, float* gdir) {
    const int ps = q.ps, x = p % ps, y = (p / ps) % ps, n = p / (ps * ps);
    const float a = alpha[p], w = patch_w[n];
    const RlV3 nfg = rl_ld(nfg_a, p), light = rl_ld(patch_light, n);
    float dot = 0.f, o[3], go[3], gi[3] = {0.f, 0.f, 0.f}, ga = 0.f, gsh = 0.f;
    const float shading = q.shaded ? rl_shading(q, nfg, light, &dot) : 1.0f;
    rl_colour(q, lut_x, lut_y, rl_ld(image, p), a, shading, o, nullptr, nullptr, nullptr, nullptr);
    for (int k = 0; k < 3; ++k) go[k] = gl * q.c_rgb * w * rl_sign(o[k] - tgt_rgb[3 * p + k]) + (g_rgb_ext ? g_rgb_ext[3 * p + k] : 0.f);
    rl_colour(q, lut_x, lut_y, rl_ld(image, p), a, shading, o, go, gi, &ga, &gsh);
    for (int k = 0; k < 3; ++k) g_image[3 * p + k] = gi[k];
    ga += gl * q.c_alpha * w * rl_sign(a - tgt_m[p]);
    {   // background bin of the entropy term: -c b (log max(b, 1e-6) - log bg_width), b = 1 - a
        const float b = 1.0f - a;
        ga += gl * q.c_ent * (logf(fmaxf(b, 1e-6f)) - q.log_bgw + (b >= 1e-6f ? 1.0f : 0.f));
    }
    RlV3 gn = rl_v(0.f, 0.f, 0.f);
    if (q.shaded && dot >= 0.f) {   // shading = clamp(dot, 0) (1 - ambient) + ambient;  n_opencv = (2 n.x - 1, -2 n.y + 1, -2 n.z + 1)
        const float gd = gsh * (1.0f - q.ambient);
        gn = rl_v(gd * 2.0f * light.x, -gd * 2.0f * light.y, -gd * 2.0f * light.z);
    }
    if (g_nrm_ext) {                // out_normals = nfg a + normal_bg (1 - a)
        const RlV3 ge = rl_ld(g_nrm_ext, p);
        gn = rl_add(gn, rl_mul(ge, a));
        ga += ge.x * (nfg.x - q.nbg[0]) + ge.y * (nfg.y - q.nbg[1]) + ge.z * (nfg.z - q.nbg[2]);
    }
    g_alpha_part[p] = ga;
    // TV: this pixel's normal enters its own term (minus sign) and the terms of the pixels above and to the left (plus sign)
    if (q.c_tv != 0.f) {
        float wh, ww, gh[3], gw[3];
        const float c = gl * q.c_tv;
        rl_tv_term(nfg_a, wfg, tgt_n, ps, n, y, x, &wh, &ww, gh, gw);
        gn = rl_sub(gn, rl_v(c * (wh * gh[0] + ww * gw[0]), c * (wh * gh[1] + ww * gw[1]), c * (wh * gh[2] + ww * gw[2])));
        if (y > 0) {
            rl_tv_term(nfg_a, wfg, tgt_n, ps, n, y - 1, x, &wh, &ww, gh, gw);
            gn = rl_add(gn, rl_v(c * wh * gh[0], c * wh * gh[1], c * wh * gh[2]));
        }
        if (x > 0) {
            rl_tv_term(nfg_a, wfg, tgt_n, ps, n, y, x - 1, &wh, &ww, gh, gw);
            gn = rl_add(gn, rl_v(c * ww * gw[0], c * ww * gw[1], c * ww * gw[2]));
        }
    }
    RlStencil gs;
    rl_normal_fg_bwd(rl_stencil(xyz, ps, n, y, x), gn, &gs);
    rl_st(gdir, 4 * p, gs.r); rl_st(gdir, 4 * p + 1, gs.u); rl_st(gdir, 4 * p + 2, gs.l); rl_st(gdir, 4 * p + 3, gs.d);
}

// ---- pass 2 (backward): gather the difference-vector gradients that touch this pixel's point, then down to depth and alpha ----------
MVE_RL_FN void rl_depth_bwd(const RlParams& q, const float* gdir, const float* g_alpha_part, const float* alpha, const float* depth,
                            const float* dir, const float* tgt_depth, const float* patch_w, float gl, int p, float* g_alpha, float* g_depth) {
    const int ps = q.ps, x = p % ps, y = (p / ps) % ps, n = p / (ps * ps), base = n * ps * ps;
    RlV3 g = rl_v(0.f, 0.f, 0.f);
    for (int xq = x - 1; xq <= x + 1; ++xq) {
        if (xq < 0 || xq >= ps) continue;
        const int pq = base + y * ps + xq;
        const int xr = xq < ps - 2 ? xq : ps - 2, xl = xq > 1 ? xq : 1;
        const float cr = (float)((xr + 1 == x) - (xr == x)), cl = (float)((xl - 1 == x) - (xl == x));
        g = rl_add(g, rl_add(rl_mul(rl_ld(gdir, 4 * pq), cr), rl_mul(rl_ld(gdir, 4 * pq + 2), cl)));
    }
    for (int yq = y - 1; yq <= y + 1; ++yq) {
        if (yq < 0 || yq >= ps) continue;
        const int pq = base + yq * ps + x;
        const int yu = yq > 1 ? yq : 1, yd = yq < ps - 2 ? yq : ps - 2;
        const float cu = (float)((yu - 1 == y) - (yu == y)), cd = (float)((yd + 1 == y) - (yd == y));
        g = rl_add(g, rl_add(rl_mul(rl_ld(gdir, 4 * pq + 1), cu), rl_mul(rl_ld(gdir, 4 * pq + 3), cd)));
    }
    const RlV3 d = rl_ld(dir, p);
    const float dn = sqrtf(rl_dot(d, d)), a = alpha[p], ac = fmaxf(a, 1e-6f), od = depth[p] * dn, dfg = od / ac, dc = fmaxf(dfg, 1e-6f);
    const float g_dfg = dfg >= 1e-6f ? -rl_dot(g, d) / (dc * dc) : 0.f;
    float g_od = g_dfg / ac;
    float ga = g_alpha_part[p];
    if (a >= 1e-6f) ga -= g_dfg * od / (ac * ac);
    if (tgt_depth) g_od += gl * q.c_depth * patch_w[n] * rl_sign(od - tgt_depth[p]);
    g_alpha[p] = ga;
    g_depth[p] = g_od * dn;
}

// ---- entropy over the sample bins (:596-603): term and gradient of one sample ---------------------------------------------------------
MVE_RL_FN float rl_entropy_sample(const RlParams& q, float w, float bin_width, float gl, float* g_w) {
    const float l = logf(fmaxf(w, 1e-6f)) - logf(fmaxf(bin_width, 1e-6f));
    if (g_w) *g_w = -gl * q.c_ent * (l + (w >= 1e-6f ? 1.0f : 0.f));
    return -q.c_ent * w * l;
}
