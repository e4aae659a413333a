// Tri-plane radiance decoders (gfx950; FMA-bound per-point MLP, gather-bound plane fetches): TriPlaneDecoder.point_decode
// (lib/models/decoders/triplane_decoder.py:135-199) and TriPlaneiNGPDecoder.point_decode (lib/models/decoders/triplane_ingp_decoder.py:
// 142-212), forward.  One wave = 64 points.  The first layer is accumulated feature by feature as the plane fetches (and hash-grid levels)
// produce them -- the 96-wide feature vector never exists; weights are wave-uniform and transposed ([in][out]), so one input's fan-out is
// one contiguous scalar load; the activated hidden vector goes through LDS ([k][lane]: conflict-free) so that the second layer can loop over
// its inputs at run time with all its outputs in registers.  The hash-grid encoding is the one of nerf.hip (hashgrid.h).
#include "raymarch_core.h"

#include "hashgrid.h"
#include "sh_core.h"

namespace {

constexpr int WAVE = 64;

struct ShK16 { float k[MVE_SH_MAX_DEGREE * MVE_SH_MAX_DEGREE]; };

struct TriParams {
    const float *xyz, *dirs, *code;
    int N, C, h, w;
    int axes[6];
    int flip_z;
    const float *base_wT, *base_b, *ingp_wT, *ingp_b, *dens_w, *dens_b, *col1_wT, *col1_b, *col2_w, *col2_b;
    int activation, sigma_activation;
    float sat;
    float *sigmas, *rgbs;
    DecoderParams hg;         // table / level meta / bound of the hash grid (only .table, .bound, .g are used)
};

__device__ __forceinline__ float act_fn(int a, float x) {
    if (a == 0) return fmaxf(x, 0.f);
    if (a == 1) return x / (1.0f + __expf(-x));
    if (a == 2) return x > 20.f ? x : log1pf(__expf(x));          // torch.nn.Softplus (beta 1, threshold 20)
    return __expf(x);                                              // trunc_exp forward
}

template <int H, int H2, int NL>
__global__ __launch_bounds__(WAVE) void k_triplane_decode(const TriParams p, const ShK16 kk) {
    __shared__ float hid[(H + 16) * WAVE];            // activated hidden vector + SH encoding, [k][lane]
    const int lane = threadIdx.x;
    const int i = blockIdx.x * WAVE + lane;
    const bool live = i < p.N;
    const int ii = live ? i : p.N - 1;
    const float x[3] = {p.xyz[3 * ii], p.xyz[3 * ii + 1], p.flip_z ? -p.xyz[3 * ii + 2] : p.xyz[3 * ii + 2]};
    float acc[H];
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = p.base_b[j];
    // ---- plane features, accumulated into the first layer as they come -----------------------------------------------------
    for (int pl = 0; pl < 3; ++pl) {
        const float u = x[p.axes[2 * pl]], v = x[p.axes[2 * pl + 1]];
        // F.grid_sample(align_corners=False, padding_mode='border'): unnormalise, clip to [0, size - 1], bilinear
        float fx = ((u + 1.0f) * (float)p.w - 1.0f) * 0.5f, fy = ((v + 1.0f) * (float)p.h - 1.0f) * 0.5f;
        fx = fminf(fmaxf(fx, 0.f), (float)(p.w - 1));
        fy = fminf(fmaxf(fy, 0.f), (float)(p.h - 1));
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = x0 + 1 < p.w ? x0 + 1 : p.w - 1, y1 = y0 + 1 < p.h ? y0 + 1 : p.h - 1;     // weight 0 where clipped
        const float tx = fx - x0f, ty = fy - y0f;
        const float w00 = (1.f - tx) * (1.f - ty), w10 = tx * (1.f - ty), w01 = (1.f - tx) * ty, w11 = tx * ty;
        const float* pc = p.code + (size_t)pl * p.h * p.w * p.C;
        const float* c00 = pc + ((size_t)y0 * p.w + x0) * p.C;
        const float* c10 = pc + ((size_t)y0 * p.w + x1) * p.C;
        const float* c01 = pc + ((size_t)y1 * p.w + x0) * p.C;
        const float* c11 = pc + ((size_t)y1 * p.w + x1) * p.C;
        for (int c = 0; c < p.C; ++c) {
            // torch's bilinear kernel sums nw, ne, sw, se in this order
            const float f = c00[c] * w00 + c10[c] * w10 + c01[c] * w01 + c11[c] * w11;
            const float* wr = p.base_wT + (size_t)(c * 3 + pl) * H;
#pragma unroll
            for (int j = 0; j < H; ++j) acc[j] = fmaf(wr[j], f, acc[j]);
        }
    }
    if constexpr (NL > 0) {
        float enc[2 * NL];
        hash_encode<NL>(p.hg, x[0], x[1], p.flip_z ? -x[2] : x[2], enc);          // the hash grid sees the un-flipped point
#pragma unroll
        for (int e = 0; e < 2 * NL; ++e) {
            const float* wr = p.ingp_wT + (size_t)e * H;
#pragma unroll
            for (int j = 0; j < H; ++j) acc[j] = fmaf(wr[j], enc[e], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < H; ++j) acc[j] += p.ingp_b[j];
    }
    // ---- density head; activated hidden vector to LDS ------------------------------------------------------------------------
    float sg = p.dens_b[0];
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const float a = act_fn(p.activation, acc[j]);
        hid[j * WAVE + lane] = a;
        sg = fmaf(p.dens_w[j], a, sg);
    }
    if (live) p.sigmas[i] = act_fn(p.sigma_activation, sg);
    if (p.dirs == nullptr) return;
    {
        float sh[16];
        she_eval(kk.k, p.dirs[3 * ii], p.dirs[3 * ii + 1], p.dirs[3 * ii + 2], 4, sh, nullptr);
#pragma unroll
        for (int k = 0; k < 16; ++k) hid[(H + k) * WAVE + lane] = sh[k];
    }
    // ---- colour net: Linear(H + 16 -> H2), act, Linear(H2 -> 3), sigmoid, saturation -------------------------------------------
    float a2[H2];
#pragma unroll
    for (int j = 0; j < H2; ++j) a2[j] = p.col1_b[j];
    for (int k = 0; k < H + 16; ++k) {
        const float a = hid[k * WAVE + lane];
        const float* wr = p.col1_wT + (size_t)k * H2;
#pragma unroll
        for (int j = 0; j < H2; ++j) a2[j] = fmaf(wr[j], a, a2[j]);
    }
    float o[3] = {p.col2_b[0], p.col2_b[1], p.col2_b[2]};
#pragma unroll
    for (int j = 0; j < H2; ++j) {
        const float a = act_fn(p.activation, a2[j]);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = fmaf(p.col2_w[c * H2 + j], a, o[c]);
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p.rgbs[3 * i + c] = (1.0f / (1.0f + __expf(-o[c]))) * (1.0f + 2.0f * p.sat) - p.sat;
    }
}

template <int H, int H2>
int launch_nl(const TriParams& p, const ShK16& kk, int n_levels, hipStream_t s) {
    const unsigned grid = mve_cdiv((unsigned)p.N, WAVE);
    switch (n_levels) {
        case 0: k_triplane_decode<H, H2, 0><<<grid, WAVE, 0, s>>>(p, kk); break;
        case 12: k_triplane_decode<H, H2, 12><<<grid, WAVE, 0, s>>>(p, kk); break;
        case 14: k_triplane_decode<H, H2, 14><<<grid, WAVE, 0, s>>>(p, kk); break;
        case 16: k_triplane_decode<H, H2, 16><<<grid, WAVE, 0, s>>>(p, kk); break;
        default: mve_set_error("triplane_decode: n_levels must be 0 (no hash grid), 12, 14 or 16 (got %d)", n_levels); return MVE_ERR_ARG;
    }
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // namespace

extern "C" int mve_triplane_decode(const MveTriplaneDesc* d, void* stream) {
    MVE_CHECK(d, MVE_ERR_ARG, "triplane_decode: null descriptor");
    if (d->N == 0) return MVE_OK;
    MVE_CHECK(d->d_xyz && d->d_code && d->d_base_wT && d->d_base_b && d->d_dens_w && d->d_dens_b && d->d_sigmas, MVE_ERR_ARG, "triplane_decode: null pointer");
    MVE_CHECK(d->N > 0 && d->C > 0 && d->h > 0 && d->w > 0, MVE_ERR_ARG, "triplane_decode: bad sizes N=%d C=%d h=%d w=%d", d->N, d->C, d->h, d->w);
    MVE_CHECK((d->hidden == 64 || d->hidden == 128) && (d->hidden2 == 64 || d->hidden2 == 128), MVE_ERR_ARG,
              "triplane_decode: hidden widths must be 64 or 128 (got %d, %d)", d->hidden, d->hidden2);
    MVE_CHECK(d->activation >= 0 && d->activation <= 2 && d->sigma_activation >= 0 && d->sigma_activation <= 3, MVE_ERR_ARG, "triplane_decode: bad activation code");
    for (int k = 0; k < 6; ++k) MVE_CHECK(d->axes[k] >= 0 && d->axes[k] < 3, MVE_ERR_ARG, "triplane_decode: bad plane axis %d", d->axes[k]);
    if (d->d_dirs) MVE_CHECK(d->d_col1_wT && d->d_col1_b && d->d_col2_w && d->d_col2_b && d->d_rgbs, MVE_ERR_ARG, "triplane_decode: null colour-net pointer");
    TriParams p;
    memset(&p, 0, sizeof(p));
    p.xyz = d->d_xyz; p.dirs = d->d_dirs; p.code = d->d_code; p.N = d->N; p.C = d->C; p.h = d->h; p.w = d->w;
    for (int k = 0; k < 6; ++k) p.axes[k] = d->axes[k];
    p.flip_z = d->flip_z;
    p.base_wT = d->d_base_wT; p.base_b = d->d_base_b; p.ingp_wT = d->d_ingp_wT; p.ingp_b = d->d_ingp_b;
    p.dens_w = d->d_dens_w; p.dens_b = d->d_dens_b; p.col1_wT = d->d_col1_wT; p.col1_b = d->d_col1_b; p.col2_w = d->d_col2_w; p.col2_b = d->d_col2_b;
    p.activation = d->activation; p.sigma_activation = d->sigma_activation; p.sat = d->sigmoid_saturation;
    p.sigmas = d->d_sigmas; p.rgbs = d->d_rgbs;
    int nl = 0;
    if (d->d_ingp_wT) {
        nl = d->n_levels;
        MVE_CHECK(d->d_ingp_b && d->d_table && d->level_scale && d->level_res && d->level_offset && d->level_size && nl > 0 && nl <= MAX_LEVELS, MVE_ERR_ARG,
                  "triplane_decode: incomplete hash-grid description");
        p.hg.table = d->d_table; p.hg.bound = d->bound; p.hg.g.n_levels = nl;
        for (int l = 0; l < nl; ++l) { p.hg.g.scale[l] = d->level_scale[l]; p.hg.g.res[l] = d->level_res[l]; p.hg.g.off[l] = d->level_offset[l]; p.hg.g.size[l] = d->level_size[l]; }
    }
    ShK16 kk;
    she_constants(kk.k);
    hipStream_t s = (hipStream_t)stream;
    if (d->hidden == 128 && d->hidden2 == 128) return launch_nl<128, 128>(p, kk, nl, s);
    if (d->hidden == 128 && d->hidden2 == 64) return launch_nl<128, 64>(p, kk, nl, s);
    if (d->hidden == 64 && d->hidden2 == 128) return launch_nl<64, 128>(p, kk, nl, s);
    return launch_nl<64, 64>(p, kk, nl, s);
}
