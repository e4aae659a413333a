// Image-space loss of one NeRF optimisation iteration and its gradients (gfx950; HBM-bound gathers, no atomics): the ~150 elementwise /
// stencil torch kernels (forward + autograd) of lib/pipelines/mvedit_3d_pipeline.py:542-603 -- depth -> normals, shading and tone
// mapping, weighted L1 on colour / alpha / depth, TV^1.5 on the normals, entropy over the sample bins -- as five forward and three
// backward launches over 8 x 128^2 rays.  All arithmetic lives in recon_loss_core.h, which the CPU tests build for the host and run
// against the reference's own statements; this file only assigns pixels to threads and reduces the loss terms (fixed-order tree per
// block, then one block over the partials: run-to-run deterministic).  Compiled without fma contraction, like the host build.
#include "common.h"
#include "reduce.h"

#include "recon_loss_core.h"
#include "mesh_loss_core.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_STEPS = 64;
constexpr int NPART = 5;   // rgb, alpha, tv, depth, entropy

struct Ws {
    float *xyz, *nfg, *wfg, *gap, *gdir, *part;
    unsigned nbp, nbs, nb;   // pixel blocks, sample blocks, partial columns (= 2 nbp + nbs)
};

size_t ws_floats(size_t N, size_t M) { return 20 * N + (size_t)NPART * (2 * mve_cdiv(N, NT) + mve_cdiv(M, NT)); }

Ws carve(void* ws, size_t N, size_t M) {
    Ws w;
    float* f = static_cast<float*>(ws);
    w.xyz = f; w.nfg = f + 3 * N; w.wfg = f + 6 * N; w.gap = f + 7 * N; w.gdir = f + 8 * N; w.part = f + 20 * N;
    w.nbp = (unsigned)mve_cdiv(N, NT); w.nbs = (unsigned)mve_cdiv(M, NT); w.nb = 2 * w.nbp + w.nbs;
    return w;
}

struct Lut { const float* x; const float* y; };

__device__ __forceinline__ void load_lut(const RlParams& q, Lut l, float* tx, float* ty) {
    if (q.lut_n > 0 && (int)threadIdx.x < q.lut_n) { tx[threadIdx.x] = l.x[threadIdx.x]; ty[threadIdx.x] = l.y[threadIdx.x]; }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void k_rl_xyz(int N, const float* __restrict__ depth, const float* __restrict__ alpha,
                                               const float* __restrict__ dir, float* __restrict__ xyz) {
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p < N) rl_st(xyz, p, rl_xyz(depth, alpha, dir, p));
}

// part layout: part[k * nb + column]; pixel pass -> columns [0, nbp), TV pass -> [nbp, 2 nbp), samples -> [2 nbp, nb)
__global__ __launch_bounds__(NT) void k_rl_pixel(RlParams q, Lut l, MveReconLossDesc d, Ws w, float* __restrict__ out_rgbs,
                                                 float* __restrict__ out_normals) {
    __shared__ float tx[MAX_STEPS], ty[MAX_STEPS], sh[NT];
    load_lut(q, l, tx, ty);
    const int N = q.P * q.ps * q.ps, p = blockIdx.x * NT + threadIdx.x;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    if (p < N)
        rl_pixel_fwd(q, tx, ty, w.xyz, d.d_image, d.d_weights_sum, d.d_depth, d.d_target_dir, d.d_target_rgbs, d.d_target_m, d.d_target_depth,
                     d.d_patch_w, d.d_patch_lights, p, w.nfg, w.wfg, out_rgbs, out_normals, part);
    const int slot[4] = {0, 1, 3, 4};
    for (int k = 0; k < 4; ++k) {
        const float s = mve_block_sum<NT>(part[k], sh);
        if (threadIdx.x == 0) w.part[slot[k] * w.nb + blockIdx.x] = s;
    }
    if (threadIdx.x == 0) w.part[2 * w.nb + blockIdx.x] = 0.f;
}

__global__ __launch_bounds__(NT) void k_rl_tv(RlParams q, const float* __restrict__ tgt_n, Ws w) {
    __shared__ float sh[NT];
    const int N = q.P * q.ps * q.ps, p = blockIdx.x * NT + threadIdx.x;
    float v = 0.f;
    if (p < N && q.c_tv != 0.f)
        v = q.c_tv * rl_tv_term(w.nfg, w.wfg, tgt_n, q.ps, p / (q.ps * q.ps), (p / q.ps) % q.ps, p % q.ps, nullptr, nullptr, nullptr, nullptr);
    const float s = mve_block_sum<NT>(v, sh);
    if (threadIdx.x == 0)
        for (int k = 0; k < NPART; ++k) w.part[k * w.nb + w.nbp + blockIdx.x] = k == 2 ? s : 0.f;
}

// forward (g_w == nullptr): entropy partials; backward: per-sample gradient
__global__ __launch_bounds__(NT) void k_rl_entropy(RlParams q, const float* __restrict__ weights, const float* __restrict__ ts, unsigned M,
                                                   const float* __restrict__ d_gl, Ws w, float* __restrict__ g_w) {
    __shared__ float sh[NT];
    const unsigned i = blockIdx.x * NT + threadIdx.x;
    const float gl = d_gl ? *d_gl : 1.0f;
    float v = 0.f;
    if (i < M) v = rl_entropy_sample(q, weights[i], ts[2 * i + 1], gl, g_w ? g_w + i : nullptr);
    if (g_w) return;
    const float s = mve_block_sum<NT>(v, sh);
    if (threadIdx.x == 0)
        for (int k = 0; k < NPART; ++k) w.part[k * w.nb + 2 * w.nbp + blockIdx.x] = k == 4 ? s : 0.f;
}

// losses[0] = total, [1..5] = rgb, alpha, tv, depth, entropy
__global__ __launch_bounds__(NT) void k_rl_reduce(Ws w, float* __restrict__ losses) {
    __shared__ float sh[NT];
    float total = 0.f;
    for (int k = 0; k < NPART; ++k) {
        float v = 0.f;
        for (unsigned c = threadIdx.x; c < w.nb; c += NT) v += w.part[k * w.nb + c];
        const float s = mve_block_sum<NT>(v, sh);
        total += s;
        if (threadIdx.x == 0) losses[1 + k] = s;
    }
    if (threadIdx.x == 0) losses[0] = total;
}

__global__ __launch_bounds__(NT) void k_rl_pixel_bwd(RlParams q, Lut l, MveReconLossDesc d, Ws w, const float* __restrict__ g_rgb_ext,
                                                     const float* __restrict__ g_nrm_ext, const float* __restrict__ d_gl,
                                                     float* __restrict__ g_image) {
    __shared__ float tx[MAX_STEPS], ty[MAX_STEPS];
    load_lut(q, l, tx, ty);
    const float gl = d_gl ? *d_gl : 1.0f;
    const int N = q.P * q.ps * q.ps, p = blockIdx.x * NT + threadIdx.x;
    if (p < N)
        rl_pixel_bwd(q, tx, ty, w.xyz, w.nfg, w.wfg, d.d_image, d.d_weights_sum, d.d_target_rgbs, d.d_target_m, d.d_target_n, d.d_patch_w,
                     d.d_patch_lights, g_rgb_ext, g_nrm_ext, gl, p, g_image, w.gap, w.gdir);
}

__global__ __launch_bounds__(NT) void k_rl_depth_bwd(RlParams q, MveReconLossDesc d, Ws w, const float* __restrict__ d_gl,
                                                     float* __restrict__ g_alpha, float* __restrict__ g_depth) {
    const float gl = d_gl ? *d_gl : 1.0f;
    const int N = q.P * q.ps * q.ps, p = blockIdx.x * NT + threadIdx.x;
    if (p < N) rl_depth_bwd(q, w.gdir, w.gap, d.d_weights_sum, d.d_depth, d.d_target_dir, d.d_target_depth, d.d_patch_w, gl, p, g_alpha, g_depth);
}

int check_desc(const MveReconLossDesc* d, const char* who) {
    MVE_CHECK(d, MVE_ERR_ARG, "%s: null descriptor", who);
    MVE_CHECK(d->P > 0 && d->ps >= 2 && (long long)d->P * d->ps * d->ps < (1ll << 27), MVE_ERR_ARG, "%s: bad patch geometry P=%d ps=%d", who, d->P, d->ps);
    MVE_CHECK(d->d_image && d->d_weights_sum && d->d_depth && d->d_target_dir && d->d_target_rgbs && d->d_target_m && d->d_patch_w &&
              d->d_patch_lights, MVE_ERR_ARG, "%s: null pointer", who);
    MVE_CHECK(d->M == 0 || (d->d_weights && d->d_ts), MVE_ERR_ARG, "%s: M = %u samples but no weights / ts", who, d->M);
    MVE_CHECK(d->lut_steps == 0 || (d->d_lut_x && d->d_lut_y && d->lut_steps >= 2 && d->lut_steps <= MAX_STEPS), MVE_ERR_ARG,
              "%s: tone-mapping table needs 2 <= steps <= %d and both arrays", who, MAX_STEPS);
    MVE_CHECK(d->bg_width > 0.f, MVE_ERR_ARG, "%s: bg_width must be positive", who);
    return MVE_OK;
}

RlParams params_of(const MveReconLossDesc* d) {
    return rl_make_params(d->P, d->ps, d->shaded ? 1 : 0, d->is_init ? 1 : 0, d->lut_steps, d->ambient_light, d->bg_color, d->normal_bg,
                          d->pixel_loss_weight, d->normal_reg_weight, d->depth_weight, d->entropy_weight, d->bg_width);
}

// ---- mesh_optim (mvedit_3d_pipeline.py:745-782): the same structure over whole rendered views ----------------------------------------
struct MWs {
    float *xyz, *craw, *cosp, *alpha, *nfg, *part;     // part[3][2 nbp]: rgb, alpha, tv
    unsigned nbp;
};

size_t mws_floats(size_t N) { return 9 * N + (size_t)3 * 2 * mve_cdiv(N, NT); }

MWs mcarve(void* ws, size_t N) {
    MWs w;
    float* f = static_cast<float*>(ws);
    w.xyz = f; w.craw = f + 3 * N; w.cosp = f + 4 * N; w.alpha = f + 5 * N; w.nfg = f + 6 * N; w.part = f + 9 * N;
    w.nbp = (unsigned)mve_cdiv(N, NT);
    return w;
}

__global__ __launch_bounds__(NT) void k_ml_xyz(int N, const float* __restrict__ depth, const float* __restrict__ dir, float* __restrict__ xyz) {
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p < N) rl_st(xyz, p, ml_xyz(depth, dir, p));
}

__global__ __launch_bounds__(NT) void k_ml_cos(int N, int S, const float* __restrict__ xyz, const float* __restrict__ dir, float* __restrict__ craw) {
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p < N) craw[p] = ml_cos_raw(xyz, dir, S, p);
}

__global__ __launch_bounds__(NT) void k_ml_pixel(MlParams q, MveMeshLossDesc d, MWs w, float* __restrict__ out_rgbs, float* __restrict__ out_normals) {
    __shared__ float sh[NT];
    const int N = q.n * q.S * q.S, p = blockIdx.x * NT + threadIdx.x;
    float part[2] = {0.f, 0.f};
    if (p < N)
        ml_pixel_fwd(q, w.craw, d.d_rgba, d.d_normal, d.d_target_rgbs, d.d_target_m_erode, d.d_target_m_blur, d.d_view_w, p, w.cosp, w.alpha, w.nfg,
                     out_rgbs, out_normals, part);
    const float s0 = mve_block_sum<NT>(part[0], sh), s1 = mve_block_sum<NT>(part[1], sh);
    if (threadIdx.x == 0) { w.part[blockIdx.x] = s0; w.part[2 * w.nbp + blockIdx.x] = s1; w.part[4 * w.nbp + blockIdx.x] = 0.f; }
}

__global__ __launch_bounds__(NT) void k_ml_tv(MlParams q, const float* __restrict__ tgt_n, MWs w) {
    __shared__ float sh[NT];
    const int N = q.n * q.S * q.S, p = blockIdx.x * NT + threadIdx.x;
    float v = 0.f;
    if (p < N && q.c_tv != 0.f)
        v = q.c_tv * rl_tv_term(w.nfg, w.alpha, tgt_n, q.S, p / (q.S * q.S), (p / q.S) % q.S, p % q.S, nullptr, nullptr, nullptr, nullptr);
    const float s = mve_block_sum<NT>(v, sh);
    if (threadIdx.x == 0) { w.part[w.nbp + blockIdx.x] = 0.f; w.part[3 * w.nbp + blockIdx.x] = 0.f; w.part[5 * w.nbp + blockIdx.x] = s; }
}

// losses[0] = total, [1..3] = rgb, alpha, tv
__global__ __launch_bounds__(NT) void k_ml_reduce(MWs w, float* __restrict__ losses) {
    __shared__ float sh[NT];
    float total = 0.f;
    for (int k = 0; k < 3; ++k) {
        float v = 0.f;
        for (unsigned c = threadIdx.x; c < 2 * w.nbp; c += NT) v += w.part[k * 2 * w.nbp + c];
        const float s = mve_block_sum<NT>(v, sh);
        total += s;
        if (threadIdx.x == 0) losses[1 + k] = s;
    }
    if (threadIdx.x == 0) losses[0] = total;
}

__global__ __launch_bounds__(NT) void k_ml_pixel_bwd(MlParams q, MveMeshLossDesc d, MWs w, const float* __restrict__ g_rgb_ext,
                                                     const float* __restrict__ g_nrm_ext, const float* __restrict__ d_gl, float* __restrict__ g_rgba,
                                                     float* __restrict__ g_normal) {
    const int N = q.n * q.S * q.S, p = blockIdx.x * NT + threadIdx.x;
    const float gl = d_gl ? *d_gl : 1.0f;
    if (p < N)
        ml_pixel_bwd(q, w.cosp, w.alpha, w.nfg, d.d_rgba, d.d_normal, d.d_target_rgbs, d.d_target_m_erode, d.d_target_m_blur, d.d_target_n,
                     d.d_view_w, g_rgb_ext, g_nrm_ext, gl, p, g_rgba, g_normal);
}

int check_mdesc(const MveMeshLossDesc* d, const char* who) {
    MVE_CHECK(d, MVE_ERR_ARG, "%s: null descriptor", who);
    MVE_CHECK(d->n > 0 && d->size >= 2 && (long long)d->n * d->size * d->size < (1ll << 27), MVE_ERR_ARG, "%s: bad view geometry n=%d size=%d", who,
              d->n, d->size);
    MVE_CHECK(d->d_rgba && d->d_normal && d->d_depth && d->d_target_dir && d->d_target_rgbs && d->d_target_m_erode && d->d_target_m_blur &&
              d->d_view_w, MVE_ERR_ARG, "%s: null pointer", who);
    return MVE_OK;
}

}  // namespace

extern "C" {

size_t mve_recon_loss_workspace_bytes(int P, int ps, uint32_t M) {
    if (P <= 0 || ps <= 0) return 0;
    return ws_floats((size_t)P * ps * ps, M) * sizeof(float);
}

int mve_recon_loss_forward(const MveReconLossDesc* d, void* d_ws, size_t ws_bytes, float* d_losses, float* d_out_rgbs, float* d_out_normals,
                           void* stream) {
    if (int rc = check_desc(d, "recon_loss_forward")) return rc;
    MVE_CHECK(d_ws && d_losses && d_out_rgbs && d_out_normals, MVE_ERR_ARG, "recon_loss_forward: null output");
    const size_t N = (size_t)d->P * d->ps * d->ps;
    MVE_CHECK(ws_bytes >= ws_floats(N, d->M) * sizeof(float), MVE_ERR_ARG, "recon_loss_forward: workspace %zu < %zu bytes", ws_bytes,
              ws_floats(N, d->M) * sizeof(float));
    const RlParams q = params_of(d);
    const Ws w = carve(d_ws, N, d->M);
    const Lut l{d->d_lut_x, d->d_lut_y};
    hipStream_t s = (hipStream_t)stream;
    k_rl_xyz<<<w.nbp, NT, 0, s>>>((int)N, d->d_depth, d->d_weights_sum, d->d_target_dir, w.xyz);
    MVE_LAUNCH_CHECK();
    k_rl_pixel<<<w.nbp, NT, 0, s>>>(q, l, *d, w, d_out_rgbs, d_out_normals);
    MVE_LAUNCH_CHECK();
    k_rl_tv<<<w.nbp, NT, 0, s>>>(q, d->d_target_n, w);
    MVE_LAUNCH_CHECK();
    if (w.nbs) {
        k_rl_entropy<<<w.nbs, NT, 0, s>>>(q, d->d_weights, d->d_ts, d->M, nullptr, w, nullptr);
        MVE_LAUNCH_CHECK();
    }
    k_rl_reduce<<<1, NT, 0, s>>>(w, d_losses);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_recon_loss_backward(const MveReconLossDesc* d, void* d_ws, size_t ws_bytes, const float* d_g_out_rgbs, const float* d_g_out_normals,
                            const float* d_g_loss, float* d_g_image, float* d_g_weights_sum, float* d_g_depth, float* d_g_weights, void* stream) {
    if (int rc = check_desc(d, "recon_loss_backward")) return rc;
    MVE_CHECK(d_ws && d_g_image && d_g_weights_sum && d_g_depth && (d->M == 0 || d_g_weights), MVE_ERR_ARG, "recon_loss_backward: null output");
    const size_t N = (size_t)d->P * d->ps * d->ps;
    MVE_CHECK(ws_bytes >= ws_floats(N, d->M) * sizeof(float), MVE_ERR_ARG, "recon_loss_backward: workspace %zu < %zu bytes", ws_bytes,
              ws_floats(N, d->M) * sizeof(float));
    const RlParams q = params_of(d);
    const Ws w = carve(d_ws, N, d->M);
    const Lut l{d->d_lut_x, d->d_lut_y};
    hipStream_t s = (hipStream_t)stream;
    k_rl_pixel_bwd<<<w.nbp, NT, 0, s>>>(q, l, *d, w, d_g_out_rgbs, d_g_out_normals, d_g_loss, d_g_image);
    MVE_LAUNCH_CHECK();
    k_rl_depth_bwd<<<w.nbp, NT, 0, s>>>(q, *d, w, d_g_loss, d_g_weights_sum, d_g_depth);
    MVE_LAUNCH_CHECK();
    if (w.nbs) {
        k_rl_entropy<<<w.nbs, NT, 0, s>>>(q, d->d_weights, d->d_ts, d->M, d_g_loss, w, d_g_weights);
        MVE_LAUNCH_CHECK();
    }
    return MVE_OK;
}

size_t mve_mesh_loss_workspace_bytes(int n, int size) {
    if (n <= 0 || size <= 0) return 0;
    return mws_floats((size_t)n * size * size) * sizeof(float);
}

int mve_mesh_loss_forward(const MveMeshLossDesc* d, void* d_ws, size_t ws_bytes, float* d_losses, float* d_out_rgbs, float* d_out_normals,
                          void* stream) {
    if (int rc = check_mdesc(d, "mesh_loss_forward")) return rc;
    MVE_CHECK(d_ws && d_losses && d_out_rgbs && d_out_normals, MVE_ERR_ARG, "mesh_loss_forward: null output");
    const size_t N = (size_t)d->n * d->size * d->size;
    MVE_CHECK(ws_bytes >= mws_floats(N) * sizeof(float), MVE_ERR_ARG, "mesh_loss_forward: workspace %zu < %zu bytes", ws_bytes,
              mws_floats(N) * sizeof(float));
    const MlParams q = ml_make_params(d->n, d->size, d->mesh_is_simplified ? 1 : 0, d->normal_bg, d->pixel_loss_weight, d->normal_reg_weight);
    const MWs w = mcarve(d_ws, N);
    hipStream_t s = (hipStream_t)stream;
    k_ml_xyz<<<w.nbp, NT, 0, s>>>((int)N, d->d_depth, d->d_target_dir, w.xyz);
    MVE_LAUNCH_CHECK();
    k_ml_cos<<<w.nbp, NT, 0, s>>>((int)N, d->size, w.xyz, d->d_target_dir, w.craw);
    MVE_LAUNCH_CHECK();
    k_ml_pixel<<<w.nbp, NT, 0, s>>>(q, *d, w, d_out_rgbs, d_out_normals);
    MVE_LAUNCH_CHECK();
    k_ml_tv<<<w.nbp, NT, 0, s>>>(q, d->d_target_n, w);
    MVE_LAUNCH_CHECK();
    k_ml_reduce<<<1, NT, 0, s>>>(w, d_losses);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_mesh_loss_backward(const MveMeshLossDesc* d, void* d_ws, size_t ws_bytes, const float* d_g_out_rgbs, const float* d_g_out_normals,
                           const float* d_g_loss, float* d_g_rgba, float* d_g_normal, void* stream) {
    if (int rc = check_mdesc(d, "mesh_loss_backward")) return rc;
    MVE_CHECK(d_ws && d_g_rgba && d_g_normal, MVE_ERR_ARG, "mesh_loss_backward: null output");
    const size_t N = (size_t)d->n * d->size * d->size;
    MVE_CHECK(ws_bytes >= mws_floats(N) * sizeof(float), MVE_ERR_ARG, "mesh_loss_backward: workspace %zu < %zu bytes", ws_bytes,
              mws_floats(N) * sizeof(float));
    const MlParams q = ml_make_params(d->n, d->size, d->mesh_is_simplified ? 1 : 0, d->normal_bg, d->pixel_loss_weight, d->normal_reg_weight);
    const MWs w = mcarve(d_ws, N);
    k_ml_pixel_bwd<<<w.nbp, NT, 0, (hipStream_t)stream>>>(q, *d, w, d_g_out_rgbs, d_g_out_normals, d_g_loss, d_g_rgba, d_g_normal);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
