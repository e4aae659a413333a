// Per-pixel arithmetic of the image-space loss of one MESH optimisation iteration and its gradients (recon_loss.hip, second half), host /
// device like recon_loss_core.h, whose vector helpers, depth_to_normal stencil and TV term it reuses.
// Reference: lib/pipelines/mvedit_3d_pipeline.py:745-782 (`mesh_optim`, from `out_alphas = render_out['rgba']...` to the regularised sum).
// Pixels are image-major: p = (n * S + y) * S + x.  Pure gathers, no atomics.
#pragma once
#include "recon_loss_core.h"

struct MlParams {
    int n, S;                    // images, render_size
    int simplified;              // mesh_is_simplified: colour term only
    float nbg[3];
    float c_rgb, c_alpha, c_tv;  // coefficients incl. the 1 / numel of the means
};

// pixel_loss = L1LossMod(loss_weight) mean over n*S*S*C, * 4.5 for colours (:758-760), * 2.0 for alpha (:765-767); TVLoss mean over
// n*3*S*S, * normal_reg_weight * 2 (:768-771)
static inline MlParams ml_make_params(int n, int S, int simplified, const float* normal_bg, float pixel_loss_weight, float normal_reg_weight) {
    MlParams q;
    const double N = (double)n * S * S;
    q.n = n; q.S = S; q.simplified = simplified;
    for (int k = 0; k < 3; ++k) q.nbg[k] = normal_bg[k];
    q.c_rgb = (float)(pixel_loss_weight * 4.5 / (3.0 * N));
    q.c_alpha = simplified ? 0.f : (float)(pixel_loss_weight * 2.0 / N);
    q.c_tv = simplified ? 0.f : (float)(normal_reg_weight * 2.0 / (3.0 * N));
    return q;
}

// pass 1: camera-space point of the rendered surface, dir / clamp(depth, 1e-6) (depth_to_normal's first line; depth is detached, :751-752)
MVE_RL_FN RlV3 ml_xyz(const float* depth, const float* dir, int p) { return rl_mul(rl_ld(dir, p), 1.0f / fmaxf(depth[p], 1e-6f)); }

// pass 2: clamp(-(n_opencv . normalize(dir)), 0) with n_opencv = depth_to_normal(..., 'opencv') * 2 - 1 (:751-755)
MVE_RL_FN float ml_cos_raw(const float* xyz, const float* dir, int S, int p) {
    const int x = p % S, y = (p / S) % S, n = p / (S * S);
    const RlStencil s = rl_stencil(xyz, S, n, y, x);
    float t;
    const RlV3 sum = rl_add(rl_add(rl_nrm(rl_cross(s.r, s.u), &t), rl_nrm(rl_cross(s.u, s.l), &t)),
                            rl_add(rl_nrm(rl_cross(s.l, s.d), &t), rl_nrm(rl_cross(s.d, s.r), &t)));
    const RlV3 nn = rl_nrm(sum, &t);
    const RlV3 ncv = rl_v((nn.x * 0.5f + 0.5f) * 2.0f - 1.0f, (nn.y * 0.5f + 0.5f) * 2.0f - 1.0f, (nn.z * 0.5f + 0.5f) * 2.0f - 1.0f);
    const RlV3 dh = rl_nrm(rl_ld(dir, p), &t);
    return fmaxf(-rl_dot(ncv, dh), 0.0f);
}

// -F.max_pool2d(-cos, 5, stride 1, padding 2): minimum over the in-image 5 x 5 neighbourhood (:756-757)
MVE_RL_FN float ml_min_pool5(const float* c, int S, int p) {
    const int x = p % S, y = (p / S) % S, n = p / (S * S);
    float m = INFINITY;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < S && xx >= 0 && xx < S) m = fminf(m, c[(n * S + yy) * S + xx]);
        }
    return m;
}

// pass 3 (forward): pooled cosine, out_normals, out_normals_fg, out_rgbs and the two L1 terms of a pixel; part[0..1] = rgb, alpha
MVE_RL_FN void ml_pixel_fwd(const MlParams& q, const float* cos_raw, const float* rgba, const float* normal, const float* tgt_rgb,
                            const float* m_erode, const float* m_blur, const float* view_w, int p, float* cosp, float* alpha_out, float* nfg_out,
                            float* out_rgbs, float* out_normals, float* part) {
    const int n = p / (q.S * q.S);
    const float c = ml_min_pool5(cos_raw, q.S, p), a = rgba[4 * p + 3], ac = fmaxf(a, 1e-3f), w = view_w[n], me = m_erode[p];
    cosp[p] = c;
    alpha_out[p] = a;
    float lrgb = 0.f;
    for (int k = 0; k < 3; ++k) {
        const float nv = normal[3 * p + k];
        const float on = nv * c + nv * (1.0f - c);                               // value of :758 (its detach only shapes the gradient)
        out_normals[3 * p + k] = on;
        nfg_out[3 * p + k] = (on - q.nbg[k] * (1.0f - a)) / ac;
        const float o = (rgba[4 * p + k] / ac) * me + tgt_rgb[3 * p + k] * (1.0f - me);
        out_rgbs[3 * p + k] = o;
        lrgb += fabsf(o - tgt_rgb[3 * p + k]);
    }
    part[0] = q.c_rgb * w * lrgb;
    part[1] = q.c_alpha * w * fabsf(a - m_blur[p]);
}

// backward of a pixel: g_rgba[4p..], g_normal[3p..]; g_rgb_ext / g_nrm_ext: optional gradients arriving at out_rgbs / out_normals
MVE_RL_FN void ml_pixel_bwd(const MlParams& q, const float* cosp, const float* alpha, const float* nfg, const float* rgba, const float* normal,
                            const float* tgt_rgb, const float* m_erode, const float* m_blur, const float* tgt_n, const float* view_w,
                            const float* g_rgb_ext, const float* g_nrm_ext, float gl, int p, float* g_rgba, float* g_normal) {
    const int S = q.S, x = p % S, y = (p / S) % S, n = p / (S * S);
    const float a = alpha[p], ac = fmaxf(a, 1e-3f), w = view_w[n], me = m_erode[p], c = cosp[p];
    float ga = gl * q.c_alpha * w * rl_sign(a - m_blur[p]);
    for (int k = 0; k < 3; ++k) {
        const float rk = rgba[4 * p + k];
        const float o = (rk / ac) * me + tgt_rgb[3 * p + k] * (1.0f - me);
        const float go = gl * q.c_rgb * w * rl_sign(o - tgt_rgb[3 * p + k]) + (g_rgb_ext ? g_rgb_ext[3 * p + k] : 0.f);
        g_rgba[4 * p + k] = go * me / ac;
        if (a >= 1e-3f) ga -= go * me * rk / (ac * ac);
    }
    RlV3 gn = rl_v(0.f, 0.f, 0.f);                                             // d loss / d out_normals_fg: TV gather as in rl_pixel_bwd
    if (q.c_tv != 0.f) {
        float wh, ww, gh[3], gw[3];
        const float ct = gl * q.c_tv;
        rl_tv_term(nfg, alpha, tgt_n, S, n, y, x, &wh, &ww, gh, gw);
        gn = rl_sub(gn, rl_v(ct * (wh * gh[0] + ww * gw[0]), ct * (wh * gh[1] + ww * gw[1]), ct * (wh * gh[2] + ww * gw[2])));
        if (y > 0) {
            rl_tv_term(nfg, alpha, tgt_n, S, n, y - 1, x, &wh, &ww, gh, gw);
            gn = rl_add(gn, rl_v(ct * wh * gh[0], ct * wh * gh[1], ct * wh * gh[2]));
        }
        if (x > 0) {
            rl_tv_term(nfg, alpha, tgt_n, S, n, y, x - 1, &wh, &ww, gh, gw);
            gn = rl_add(gn, rl_v(ct * ww * gw[0], ct * ww * gw[1], ct * ww * gw[2]));
        }
    }
    const float gnf[3] = {gn.x, gn.y, gn.z};
    for (int k = 0; k < 3; ++k) {
        // out_normals_fg = (out_normals - nbg (1 - a)) / clamp(a, 1e-3);  out_normals = normal cos + normal.detach() (1 - cos)
        const float nv = normal[3 * p + k], on = nv * c + nv * (1.0f - c);
        ga += gnf[k] * q.nbg[k] / ac;
        if (a >= 1e-3f) ga -= gnf[k] * (on - q.nbg[k] * (1.0f - a)) / (ac * ac);
        g_normal[3 * p + k] = c * (gnf[k] / ac + (g_nrm_ext ? g_nrm_ext[3 * p + k] : 0.f));
    }
    g_rgba[4 * p + 3] = ga;
}
