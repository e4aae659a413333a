// TEST INFRASTRUCTURE ONLY.  Host build of the product's host/device "core" headers (mvedit_amd/csrc/raster_grad_core.h,
// shading_core.h): the per-pixel bodies of kernels whose first GPU run is still pending are wrapped in plain loops here, so that the
// CPU tests can run that very arithmetic against the Python closed forms, finite differences of the C oracle, and outputs of the
// reference's Tonemapping class.  Built by oracle.build_devcore() with g++ -ffp-contract=off into oracle/libdevcore.so.
#include <stddef.h>
#include <stdint.h>

#include "../mvedit_amd/csrc/raster_grad_core.h"
#include "../mvedit_amd/csrc/shading_core.h"
#include "../mvedit_amd/csrc/recon_loss_core.h"
#include "../mvedit_amd/csrc/mesh_reg_core.h"
#include "../mvedit_amd/csrc/mesh_loss_core.h"
#include "../mvedit_amd/csrc/blur_core.h"
#include "../mvedit_amd/csrc/sh_core.h"
#include <vector>

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

void dc_tonemap_lut_grad(const float* x, size_t n, const float* lut_x, const float* lut_y, int steps, int inverse, int linear, float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = sh_lut_grad(lut_x, lut_y, steps, x[i], inverse, linear);
}

void dc_shade_points(const float* albedo, const float* normal, const float* lights, size_t N, float ambient, const float* lut_x, const float* lut_y,
                     int steps, float* out, const float* g_out, float* g_albedo, float* g_normal) {
    for (size_t i = 0; i < N; ++i)
        sh_shade_point(albedo + 3 * i, normal + 3 * i, lights + 3 * i, ambient, lut_x, lut_y, lut_x ? steps : 0, out ? out + 3 * i : nullptr,
                       g_out ? g_out + 3 * i : nullptr, g_albedo ? g_albedo + 3 * i : nullptr, g_normal ? g_normal + 3 * i : nullptr);
}

void dc_shade_views(const float* rgba, const float* normal_fg, const float* lights, unsigned n_views, unsigned pix, float ambient, float bg,
                    const float* lut_x, const float* lut_y, int steps, float* image) {
    for (size_t i = 0; i < (size_t)n_views * pix; ++i)
        sh_shade_pixel(rgba + i * 4, normal_fg + i * 3, lights + 3 * (i / pix), ambient, bg, lut_x, lut_y, lut_x ? steps : 0, image + i * 3);
}

// The whole image-space loss of one NeRF optimisation iteration, forward and backward, as recon_loss.hip sequences it: xyz pass, pixel
// pass, TV pass, entropy pass; then pixel backward, depth backward, entropy backward.  losses[6] = total, rgb, alpha, tv, depth, entropy
// (summed in double here; the device reduces per block in float).  ws: 3N (xyz) + 3N (nfg) + N (wfg) + N (g_alpha_part) + 12N (gdir) floats.
void dc_recon_loss(int P, int ps, int shaded, int is_init, int lut_n, float ambient, float bg, const float* normal_bg, float pixel_loss_weight,
                   float normal_reg_weight, float depth_weight, float entropy_weight, float bg_width, const float* lut_x, const float* lut_y,
                   const float* image, const float* alpha, const float* depth, const float* weights, const float* ts, int M, const float* dir,
                   const float* tgt_rgb, const float* tgt_m, const float* tgt_n, const float* tgt_depth, const float* patch_w,
                   const float* patch_light, const float* g_rgb_ext, const float* g_nrm_ext, float gl, float* ws, double* losses,
                   float* out_rgbs, float* out_normals, float* g_image, float* g_alpha, float* g_depth, float* g_weights) {
    const RlParams q = rl_make_params(P, ps, shaded, is_init, lut_n, ambient, bg, normal_bg, pixel_loss_weight, normal_reg_weight, depth_weight,
                                      entropy_weight, bg_width);
    const int N = P * ps * ps;
    float *xyz = ws, *nfg = ws + 3 * (size_t)N, *wfg = ws + 6 * (size_t)N, *gap = ws + 7 * (size_t)N, *gdir = ws + 8 * (size_t)N;
    double sums[5] = {0, 0, 0, 0, 0};
    for (int p = 0; p < N; ++p) rl_st(xyz, p, rl_xyz(depth, alpha, dir, p));
    for (int p = 0; p < N; ++p) {
        float part[4];
        rl_pixel_fwd(q, lut_x, lut_y, xyz, image, alpha, depth, dir, tgt_rgb, tgt_m, tgt_depth, patch_w, patch_light, p, nfg, wfg, out_rgbs,
                     out_normals, part);
        sums[0] += part[0]; sums[1] += part[1]; sums[3] += part[2]; sums[4] += part[3];
    }
    if (q.c_tv != 0.f)
        for (int p = 0; p < N; ++p)
            sums[2] += q.c_tv * rl_tv_term(nfg, wfg, tgt_n, ps, p / (ps * ps), (p / ps) % ps, p % ps, nullptr, nullptr, nullptr, nullptr);
    for (int i = 0; i < M; ++i) sums[4] += rl_entropy_sample(q, weights[i], ts[2 * i + 1], gl, g_weights + i);
    losses[0] = sums[0] + sums[1] + sums[2] + sums[3] + sums[4];
    for (int k = 0; k < 5; ++k) losses[1 + k] = sums[k];
    for (int p = 0; p < N; ++p)
        rl_pixel_bwd(q, lut_x, lut_y, xyz, nfg, wfg, image, alpha, tgt_rgb, tgt_m, tgt_n, patch_w, patch_light, g_rgb_ext, g_nrm_ext, gl, p, g_image,
                     gap, gdir);
    for (int p = 0; p < N; ++p) rl_depth_bwd(q, gdir, gap, alpha, depth, dir, tgt_depth, patch_w, gl, p, g_alpha, g_depth);
}

// Both mesh regularisers, forward and backward, as mesh_reg.hip sequences them: count -> scan -> scatter the half-edges into per-vertex
// buckets -> per-vertex sort + forward -> reduce; then the two per-vertex backward passes.  losses[2] = laplacian_smooth_loss,
// normal_consistency; n_edges_out = E.  g_fn must come zero-initialised.
void dc_mesh_reg(const float* verts, int V, const int32_t* faces, int F, const float* face_normals, float gl_lap, float gl_nc, double* losses,
                 int* n_edges_out, float* g_verts, float* g_fn) {
    std::vector<int> cnt(V, 0), base(V + 1, 0), fill(V, 0);
    for (int t = 0; t < F; ++t)
        for (int k = 0; k < 3; ++k) cnt[faces[3 * t + k]] += 2;
    for (int i = 0; i < V; ++i) base[i + 1] = base[i] + cnt[i];
    std::vector<uint64_t> bucket((size_t)6 * F);
    for (int t = 0; t < F; ++t)
        for (int k = 0; k < 3; ++k) {
            const int a = faces[3 * t + k], b = faces[3 * t + (k + 1) % 3], side = a > b ? 1 : 0;
            bucket[base[a] + fill[a]++] = mr_pack(b, t, side);
            bucket[base[b] + fill[b]++] = mr_pack(a, t, side);
        }
    std::vector<float> u((size_t)3 * V);
    double lap = 0, nc = 0;
    long long E = 0;
    for (int i = 0; i < V; ++i) {
        mr_sort(bucket.data() + base[i], cnt[i]);
        float ncs; int ne;
        lap += mr_vertex_fwd(i, bucket.data() + base[i], cnt[i], verts, face_normals, u.data() + 3 * i, &ncs, &ne);
        nc += ncs; E += ne;
    }
    losses[0] = lap / V; losses[1] = E ? nc / (double)E : 0.0;
    *n_edges_out = (int)E;
    for (int i = 0; i < V; ++i) {
        mr_vertex_bwd_lap(i, bucket.data() + base[i], cnt[i], u.data(), gl_lap / (float)V, g_verts);
        if (E) mr_vertex_bwd_nc(i, bucket.data() + base[i], cnt[i], face_normals, gl_nc / (float)E, g_fn);
    }
}

// Image-space loss of one mesh optimisation iteration, forward and backward, in recon_loss.hip's launch order: xyz, raw cosine, pixel
// pass, TV pass; one backward pass.  losses[4] = total, rgb, alpha, tv.  ws: 3N (xyz) + N (cos_raw) + N (cos) + N (alpha) + 3N (nfg) floats.
void dc_mesh_loss(int n, int S, int simplified, const float* normal_bg, float pixel_loss_weight, float normal_reg_weight, const float* rgba,
                  const float* normal, const float* depth, const float* dir, const float* tgt_rgb, const float* m_erode, const float* m_blur,
                  const float* tgt_n, const float* view_w, const float* g_rgb_ext, const float* g_nrm_ext, float gl, float* ws, double* losses,
                  float* out_rgbs, float* out_normals, float* g_rgba, float* g_normal) {
    const MlParams q = ml_make_params(n, S, simplified, normal_bg, pixel_loss_weight, normal_reg_weight);
    const int N = n * S * S;
    float *xyz = ws, *craw = ws + 3 * (size_t)N, *cosp = ws + 4 * (size_t)N, *alpha = ws + 5 * (size_t)N, *nfg = ws + 6 * (size_t)N;
    double sums[3] = {0, 0, 0};
    for (int p = 0; p < N; ++p) rl_st(xyz, p, ml_xyz(depth, dir, p));
    for (int p = 0; p < N; ++p) craw[p] = ml_cos_raw(xyz, dir, S, p);
    for (int p = 0; p < N; ++p) {
        float part[2];
        ml_pixel_fwd(q, craw, rgba, normal, tgt_rgb, m_erode, m_blur, view_w, p, cosp, alpha, nfg, out_rgbs, out_normals, part);
        sums[0] += part[0]; sums[1] += part[1];
    }
    if (q.c_tv != 0.f)
        for (int p = 0; p < N; ++p)
            sums[2] += q.c_tv * rl_tv_term(nfg, alpha, tgt_n, S, p / (S * S), (p / S) % S, p % S, nullptr, nullptr, nullptr, nullptr);
    losses[0] = sums[0] + sums[1] + sums[2];
    for (int k = 0; k < 3; ++k) losses[1 + k] = sums[k];
    for (int p = 0; p < N; ++p)
        ml_pixel_bwd(q, cosp, alpha, nfg, rgba, normal, tgt_rgb, m_erode, m_blur, tgt_n, view_w, g_rgb_ext, g_nrm_ext, gl, p, g_rgba, g_normal);
}

// Mesh.auto_normal forward + backward in mesh_reg.hip's launch order.  vn_sum / g_verts must come zero-initialised; g_sum is [V,3] scratch.
void dc_mesh_normals(const float* verts, int V, const int32_t* faces, int F, const float* g_vn, const float* g_fn_ext, float* face_normals,
                     float* vn_sum, float* vn, float* g_sum, float* g_verts) {
    for (int t = 0; t < F; ++t) mr_normals_face_fwd(verts, faces, t, face_normals, vn_sum);
    for (int i = 0; i < V; ++i) {
        const float l = sqrtf(mr_dot3(vn_sum + 3 * i, vn_sum + 3 * i)), inv = 1.0f / fmaxf(l, 1e-12f);
        for (int k = 0; k < 3; ++k) vn[3 * i + k] = vn_sum[3 * i + k] * inv;
    }
    for (int i = 0; i < V; ++i) {
        const float zero[3] = {0.f, 0.f, 0.f};
        mr_normalize_bwd(vn_sum + 3 * i, g_vn ? g_vn + 3 * i : zero, g_sum + 3 * i);
    }
    for (int t = 0; t < F; ++t) mr_normals_face_bwd(verts, faces, t, g_fn_ext, g_sum, g_verts);
}

// gaussian blur of `planes` images [H, W] (rows then columns, as blur.hip), or its adjoint; base != NULL: out = offset + base - blur (highpass)
void dc_gaussian_blur(const float* x, int planes, int H, int W, int ksize, float sigma, int adjoint, const float* base, float offset, float* out) {
    std::vector<float> w(ksize), tmp((size_t)H * W);
    bl_kernel1d(ksize, sigma, w.data());
    const int r = ksize / 2;
    for (int p = 0; p < planes; ++p) {
        const float* xp = x + (size_t)p * H * W;
        for (int y = 0; y < H; ++y)
            for (int c = 0; c < W; ++c)
                tmp[(size_t)y * W + c] = adjoint ? bl_adj1d(xp + (size_t)y * W, W, 1, w.data(), r, c) : bl_fwd1d(xp + (size_t)y * W, W, 1, w.data(), r, c);
        for (int y = 0; y < H; ++y)
            for (int c = 0; c < W; ++c) {
                const float b = adjoint ? bl_adj1d(tmp.data() + c, H, W, w.data(), r, y) : bl_fwd1d(tmp.data() + c, H, W, w.data(), r, y);
                const size_t i = (size_t)p * H * W + (size_t)y * W + c;
                out[i] = base ? offset + base[i] - b : b;
            }
    }
}

void dc_sh_encode(const float* xyz, int B, int degree, float* out, float* jac) {
    float k[MVE_SH_MAX_DEGREE * MVE_SH_MAX_DEGREE];
    she_constants(k);
    const int C2 = degree * degree;
    for (int b = 0; b < B; ++b) she_eval(k, xyz[3 * b], xyz[3 * b + 1], xyz[3 * b + 2], degree, out + (size_t)b * C2, jac ? jac + (size_t)b * 3 * C2 : nullptr);
}

}  // extern "C"
