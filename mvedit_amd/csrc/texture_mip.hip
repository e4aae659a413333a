// Mip-mapped texture path of the mesh renderer (gfx950; every kernel is an HBM / gather-latency bound per-pixel pass):
//   mve_rasterize_db      dr.rasterize(...)[1]: barycentric pixel differentials of every covered pixel
//   mve_interpolate_da    dr.interpolate(attr, rast, tri, rast_db=..., diff_attrs='all')[1]
//   mve_mip_build         the box-filtered level stack nvdiffrast builds inside dr.texture
//   mve_texture_mip       dr.texture(tex, uv, uv_da=..., filter_mode='linear-mipmap-linear') (wrap addressing), background -> 0 with rast
//   mve_texture_mip_backward   its gradient w.r.t. the texture: trilinear weights scattered into the level stack (float atomics, as
//                         nvdiffrast does), then gathered down to level 0 per texel (deterministic); FIXED = the 2^-32 fixed-point variant
//                         that get_cam_weights_uv / bake_multiview use for `visibility_grad` (base_mesh_renderer.py:470-475, :547-552)
//   mve_bake_accumulate_mip    the per-texel multi-view gather of bake_multiview with the mip-mapped image fetch (:566-582)
// Reference call sites: lib/models/decoders/mesh_renderer/base_mesh_renderer.py:196, :241, :260-264, :357-361, :442, :466-474, :496-500,
// :544-551, :573-577.  Arithmetic in texmip_core.h (host/device; the CPU tests run a host build against oracle/texture_mip_oracle.py).
#include "common.h"

#include "texmip_core.h"

namespace {

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void k_rasterize_db(const float* __restrict__ pos, int V, const int32_t* __restrict__ tri, int F,
                                                     const float* __restrict__ rast, int B, int H, int W, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * H * W) return;
    const f32x4 r = reinterpret_cast<const f32x4*>(rast)[i];
    const int id = (int)r[3] - 1;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (id >= 0 && id < F) {
        const size_t b = i / ((size_t)H * W);
        const int pix = (int)(i - b * (size_t)H * W), py = pix / W, px = pix - py * W;
        const float* pb = pos + b * (size_t)V * 4;
        float d[4];
        tm_rast_db(pb + 4 * (size_t)tri[3 * id], pb + 4 * (size_t)tri[3 * id + 1], pb + 4 * (size_t)tri[3 * id + 2], r[0], r[1], px, py, W, H, d);
        o = f32x4{d[0], d[1], d[2], d[3]};
    }
    reinterpret_cast<f32x4*>(out)[i] = o;
}

__global__ __launch_bounds__(NT) void k_interpolate_da(const float* __restrict__ attr, size_t attr_stride, int C, const float* __restrict__ rast,
                                                       const float* __restrict__ rast_db, size_t total, size_t npix,
                                                       const int32_t* __restrict__ tri, int F, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const f32x4 r = reinterpret_cast<const f32x4*>(rast)[i];
    const f32x4 db4 = reinterpret_cast<const f32x4*>(rast_db)[i];
    const float db[4] = {db4[0], db4[1], db4[2], db4[3]};
    const int id = (int)r[3] - 1;
    float* o = out + i * 2 * C;
    if (id < 0 || id >= F) {
        for (int c = 0; c < 2 * C; ++c) o[c] = 0.f;
        return;
    }
    const float* a = attr + (i / npix) * attr_stride;
    const float* a0 = a + (size_t)tri[3 * id] * C;
    const float* a1 = a + (size_t)tri[3 * id + 1] * C;
    const float* a2 = a + (size_t)tri[3 * id + 2] * C;
    for (int c = 0; c < C; ++c) tm_attr_da(a0[c], a1[c], a2[c], db, o + 2 * c, o + 2 * c + 1);
}

// one level: dst (h2 x w2) = box filter of src (h x w); a dimension of size 1 stays
__global__ __launch_bounds__(NT) void k_mip_build(const float* __restrict__ src, size_t src_stride, float* __restrict__ dst, size_t dst_stride,
                                                  int Bt, int h, int w, int C) {
    const int h2 = h > 1 ? h >> 1 : 1, w2 = w > 1 ? w >> 1 : 1;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)Bt * h2 * w2 * C) return;
    const int c = (int)(i % C);
    size_t t = i / C;
    const int x = (int)(t % w2); t /= w2;
    const int y = (int)(t % h2); t /= h2;
    const float* s = src + t * src_stride;
    const int y0 = h > 1 ? 2 * y : 0, y1 = h > 1 ? 2 * y + 1 : 0, x0 = w > 1 ? 2 * x : 0, x1 = w > 1 ? 2 * x + 1 : 0;
    const float top = 0.5f * (s[((size_t)y0 * w + x0) * C + c] + s[((size_t)y1 * w + x0) * C + c]);      // rows first, then columns (oracle order)
    const float bot = 0.5f * (s[((size_t)y0 * w + x1) * C + c] + s[((size_t)y1 * w + x1) * C + c]);
    dst[t * dst_stride + ((size_t)y * w2 + x) * C + c] = 0.5f * (top + bot);
}

// up to five levels per launch: a block owns a tile of <= 32 x 32 texels of level l (th x tw of them; every extent on the way down is even or 1,
// check_tex) and derives levels l + 1 .. l + nl of it through LDS, each from the STORED fp32 values of the one below, rows first then columns --
// bit-identical to nl launches of k_mip_build.  src: level l ([Bt] textures src_stride apart), mips: the stack (level l + k at offs[k - 1] * C)
struct MipTileArgs { int off[5]; };
__global__ __launch_bounds__(NT) void k_mip_build_tile(const float* __restrict__ src, size_t src_stride, float* __restrict__ mips, size_t mip_stride,
                                                       int h, int w, int C, int th, int tw, int nl, MipTileArgs a) {
    extern __shared__ float tile[];                       // two buffers of th * tw * C and th * tw * C / 2 floats
    const int tiles_x = w / tw, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const size_t t = blockIdx.y;
    float* cur = tile;
    float* nxt = tile + (size_t)th * tw * C;
    const float* s = src + t * src_stride;
    for (int i = threadIdx.x; i < th * tw * C; i += NT) {
        const int c = i % C, x = (i / C) % tw, y = i / (C * tw);
        cur[i] = s[((size_t)(ty * th + y) * w + tx * tw + x) * C + c];
    }
    __syncthreads();
    int ch = th, cw = tw, lh = h, lw = w;                 // extents of the tile / of the whole level held in `cur`
    for (int k = 0; k < nl; ++k) {
        const int nh = ch > 1 ? ch >> 1 : 1, nw = cw > 1 ? cw >> 1 : 1;
        lh = lh > 1 ? lh >> 1 : 1; lw = lw > 1 ? lw >> 1 : 1;
        float* dst = mips + t * mip_stride + (size_t)a.off[k] * C;
        for (int i = threadIdx.x; i < nh * nw * C; i += NT) {
            const int c = i % C, x = (i / C) % nw, y = i / (C * nw);
            const int y0 = ch > 1 ? 2 * y : 0, y1 = ch > 1 ? 2 * y + 1 : 0, x0 = cw > 1 ? 2 * x : 0, x1 = cw > 1 ? 2 * x + 1 : 0;
            const float top = 0.5f * (cur[(y0 * cw + x0) * C + c] + cur[(y1 * cw + x0) * C + c]);
            const float bot = 0.5f * (cur[(y0 * cw + x1) * C + c] + cur[(y1 * cw + x1) * C + c]);
            const float v = 0.5f * (top + bot);
            nxt[i] = v;
            dst[((size_t)(ty * nh + y) * lw + tx * nw + x) * C + c] = v;
        }
        __syncthreads();
        float* sw = cur; cur = nxt; nxt = sw;
        ch = nh; cw = nw;
    }
}

struct TexDesc {
    const float* tex0; const float* mips;       // [Bt][H*W*C], [Bt][mip_texels*C]
    size_t tex_stride, mip_stride;              // floats per texture (0: one texture shared by every sample)
    int H, W, C, max_level;
};

// pointer + extent of level l of the texture that sample n reads
__device__ __forceinline__ const float* level_ptr(const TexDesc& t, size_t n, int l, int* w, int* h) {
    *w = tm_dim(t.W, l); *h = tm_dim(t.H, l);
    if (l == 0) return t.tex0 + n * t.tex_stride;
    return t.mips + n * t.mip_stride + (size_t)tm_mip_offset(t.H, t.W, l) * t.C;
}

// trilinear fetch of up to 4 channels per call chunk; out[c] for c in [0, C)
__device__ __forceinline__ void mip_fetch(const TexDesc& t, size_t n, float u, float v, const float* da, float* out) {
    const TmLevel L = tm_level(da, t.W, t.H, t.max_level);
    for (int c = 0; c < t.C; ++c) out[c] = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        const int l = pass ? L.l1 : L.l0;
        const float wl = pass ? L.f : 1.0f - L.f;
        if (pass && !(L.f > 0.f)) break;
        int w, h, ix[2], iy[2];
        float fu, fv;
        const float* p = level_ptr(t, n, l, &w, &h);
        tm_taps(u, v, w, h, ix, iy, &fu, &fv);
        for (int c = 0; c < t.C; ++c) {
            const float a00 = p[((size_t)iy[0] * w + ix[0]) * t.C + c], a10 = p[((size_t)iy[0] * w + ix[1]) * t.C + c];
            const float a01 = p[((size_t)iy[1] * w + ix[0]) * t.C + c], a11 = p[((size_t)iy[1] * w + ix[1]) * t.C + c];
            const float top = a00 + fu * (a10 - a00), bot = a01 + fu * (a11 - a01);
            out[c] += wl * (top + fv * (bot - top));
        }
    }
}

constexpr int TM_MAX_C = 8;

__global__ __launch_bounds__(NT) void k_texture_mip(TexDesc t, const float* __restrict__ uv, const float* __restrict__ uv_da,
                                                    const float* __restrict__ rast, size_t total, size_t npix, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    float* o = out + i * t.C;
    if (rast && !(rast[4 * i + 3] > 0.f)) {
        for (int c = 0; c < t.C; ++c) o[c] = 0.f;
        return;
    }
    float res[TM_MAX_C];
    const f32x4 d4 = reinterpret_cast<const f32x4*>(uv_da)[i];
    const float da[4] = {d4[0], d4[1], d4[2], d4[3]};
    mip_fetch(t, i / npix, uv[2 * i], uv[2 * i + 1], da, res);
    for (int c = 0; c < t.C; ++c) o[c] = res[c];
}

// gradient scatter: g_tex0 / g_mips laid out like the texture; FIXED: C = 1, g_out = 1 on covered pixels, accumulation in 2^-32 fixed point
template <bool FIXED>
__global__ __launch_bounds__(NT) void k_texture_mip_bwd(TexDesc t, const float* __restrict__ g_out, const float* __restrict__ uv,
                                                        const float* __restrict__ uv_da, const float* __restrict__ rast, size_t total,
                                                        size_t npix, void* __restrict__ g_tex0, void* __restrict__ g_mips) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    if (rast && !(rast[4 * i + 3] > 0.f)) return;
    const size_t n = i / npix;
    const f32x4 d4 = reinterpret_cast<const f32x4*>(uv_da)[i];
    const float da[4] = {d4[0], d4[1], d4[2], d4[3]};
    const TmLevel L = tm_level(da, t.W, t.H, t.max_level);
    for (int pass = 0; pass < 2; ++pass) {
        const int l = pass ? L.l1 : L.l0;
        const float wl = pass ? L.f : 1.0f - L.f;
        if (pass && !(L.f > 0.f)) break;
        const int w = tm_dim(t.W, l), h = tm_dim(t.H, l);
        int ix[2], iy[2];
        float fu, fv;
        tm_taps(uv[2 * i], uv[2 * i + 1], w, h, ix, iy, &fu, &fv);
        const size_t base = l == 0 ? n * t.tex_stride : n * t.mip_stride + (size_t)tm_mip_offset(t.H, t.W, l) * t.C;
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                const float wt = wl * (k ? fu : 1.0f - fu) * (j ? fv : 1.0f - fv);
                const size_t e = base + ((size_t)iy[j] * w + ix[k]) * t.C;
                if constexpr (FIXED) {
                    unsigned long long* g = reinterpret_cast<unsigned long long*>(l == 0 ? g_tex0 : g_mips);
                    atomicAdd(g + e, (unsigned long long)llrint((double)wt * 4294967296.0));
                } else {
                    float* g = reinterpret_cast<float*>(l == 0 ? g_tex0 : g_mips);
                    for (int c = 0; c < t.C; ++c) atomicAdd(g + e + c, wt * g_out[i * t.C + c]);
                }
            }
    }
}

// level 0 += sum over the levels above of the covering texel's gradient / (texels of level 0 it averages); one thread per level-0 texel
template <bool FIXED>
__global__ __launch_bounds__(NT) void k_mip_collapse(TexDesc t, int Bt, const void* __restrict__ g_mips, void* __restrict__ g_tex0, float* __restrict__ out_f32) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)Bt * t.H * t.W * t.C) return;
    const int c = (int)(i % t.C);
    size_t r = i / t.C;
    const int x = (int)(r % t.W); r /= t.W;
    const int y = (int)(r % t.H); r /= t.H;
    double acc = 0.0;
    if constexpr (FIXED) acc = (double)reinterpret_cast<const unsigned long long*>(g_tex0)[i] * (1.0 / 4294967296.0);
    else acc = (double)reinterpret_cast<const float*>(g_tex0)[i];
    for (int l = 1; l <= t.max_level; ++l) {
        const int w = tm_dim(t.W, l), h = tm_dim(t.H, l);
        const int xl = t.W > 1 ? (x >> l < w ? x >> l : w - 1) : 0, yl = t.H > 1 ? (y >> l < h ? y >> l : h - 1) : 0;
        const size_t e = r * t.mip_stride + ((size_t)tm_mip_offset(t.H, t.W, l) + (size_t)yl * w + xl) * t.C + c;
        const double area = (double)(t.H / h) * (double)(t.W / w);
        if constexpr (FIXED) acc += (double)reinterpret_cast<const unsigned long long*>(g_mips)[e] * (1.0 / 4294967296.0) / area;
        else acc += (double)reinterpret_cast<const float*>(g_mips)[e] / area;
    }
    if constexpr (FIXED) out_f32[i] = (float)acc;
    else reinterpret_cast<float*>(g_tex0)[i] = (float)acc;
}

// the same for even H, W: one thread per 2 x 2 quad of level 0 -- the levels above are shared by the quad, read once, and added to each of its
// four texels in the order of k_mip_collapse (bit-identical): 2.75 reads per texel instead of 1 + max_level
constexpr int QUAD_MAX_LEVEL = 14;
template <bool FIXED>
__global__ __launch_bounds__(NT) void k_mip_collapse_quad(TexDesc t, int Bt, const void* __restrict__ g_mips, void* __restrict__ g_tex0, float* __restrict__ out_f32) {
    const int qw = t.W >> 1, qh = t.H >> 1;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)Bt * qh * qw) return;
    const int qx = (int)(i % qw), qy = (int)((i / qw) % qh);
    const size_t r = i / ((size_t)qw * qh);
    for (int c = 0; c < t.C; ++c) {
        double up[QUAD_MAX_LEVEL + 1];
#pragma unroll
        for (int l = 1; l <= QUAD_MAX_LEVEL; ++l) {
            up[l] = 0.0;
            if (l > t.max_level) continue;
            const int w = tm_dim(t.W, l), h = tm_dim(t.H, l);
            const int x = 2 * qx, y = 2 * qy;
            const int xl = x >> l < w ? x >> l : w - 1, yl = y >> l < h ? y >> l : h - 1;
            const size_t e = r * t.mip_stride + ((size_t)tm_mip_offset(t.H, t.W, l) + (size_t)yl * w + xl) * t.C + c;
            const double area = (double)(t.H / h) * (double)(t.W / w);
            if constexpr (FIXED) up[l] = (double)reinterpret_cast<const unsigned long long*>(g_mips)[e] * (1.0 / 4294967296.0) / area;
            else up[l] = (double)reinterpret_cast<const float*>(g_mips)[e] / area;
        }
        for (int j = 0; j < 4; ++j) {
            const size_t e0 = ((r * t.H + 2 * qy + (j >> 1)) * t.W + 2 * qx + (j & 1)) * t.C + c;
            double acc;
            if constexpr (FIXED) acc = (double)reinterpret_cast<const unsigned long long*>(g_tex0)[e0] * (1.0 / 4294967296.0);
            else acc = (double)reinterpret_cast<const float*>(g_tex0)[e0];
#pragma unroll
            for (int l = 1; l <= QUAD_MAX_LEVEL; ++l)
                if (l <= t.max_level) acc += up[l];
            if constexpr (FIXED) out_f32[e0] = (float)acc;
            else reinterpret_cast<float*>(g_tex0)[e0] = (float)acc;
        }
    }
}

template <bool FIXED>
void launch_collapse(const TexDesc& t, int Bt, const void* g_mips, void* g_tex0, float* out_f32, hipStream_t s) {
    if (t.H % 2 == 0 && t.W % 2 == 0 && t.max_level >= 1 && t.max_level <= QUAD_MAX_LEVEL)
        k_mip_collapse_quad<FIXED><<<mve_cdiv((size_t)Bt * (t.H / 2) * (t.W / 2), NT), NT, 0, s>>>(t, Bt, g_mips, g_tex0, out_f32);
    else
        k_mip_collapse<FIXED><<<mve_cdiv((size_t)Bt * t.H * t.W * t.C, NT), NT, 0, s>>>(t, Bt, g_mips, g_tex0, out_f32);
}

// per texel of the atlas: for every view of the batch the mip-mapped fetch of (r, g, b, view weight) at the texel's projection, times
// the texel's visibility; accum += (rgb * weight, weight)      (base_mesh_renderer.py:566-582)
__global__ __launch_bounds__(NT) void k_bake_accumulate_mip(const float* __restrict__ tex_rast, const float* __restrict__ tex_rast_db,
                                                            const int32_t* __restrict__ f, int F, const float* __restrict__ v_img, int V,
                                                            TexDesc img /* C = 4, Bt = n */, const float* __restrict__ vis, int n, int map,
                                                            float* __restrict__ accum) {
    const size_t t = (size_t)blockIdx.x * NT + threadIdx.x;
    if (t >= (size_t)map * map) return;
    const f32x4 r = reinterpret_cast<const f32x4*>(tex_rast)[t];
    const f32x4 db4 = reinterpret_cast<const f32x4*>(tex_rast_db)[t];
    const float db[4] = {db4[0], db4[1], db4[2], db4[3]};
    const int id = (int)r[3] - 1;
    f32x4 acc = reinterpret_cast<f32x4*>(accum)[t];
    for (int v = 0; v < n; ++v) {
        const float vz = vis[(size_t)v * map * map + t];
        if (vz == 0.f) continue;                                            // weight = fetch * 0: adds exactly nothing (finite images)
        float cu = 0.f, cv = 0.f, da[4] = {0.f, 0.f, 0.f, 0.f};            // dr.interpolate gives 0 on empty texels
        if (id >= 0 && id < F) {
            const float* vi = v_img + (size_t)v * V * 2;
            const int i0 = f[3 * id], i1 = f[3 * id + 1], i2 = f[3 * id + 2];
            const float bw = 1.0f - r[0] - r[1];
            cu = r[0] * vi[2 * i0] + r[1] * vi[2 * i1] + bw * vi[2 * i2];
            cv = r[0] * vi[2 * i0 + 1] + r[1] * vi[2 * i1 + 1] + bw * vi[2 * i2 + 1];
            tm_attr_da(vi[2 * i0], vi[2 * i1], vi[2 * i2], db, &da[0], &da[1]);
            tm_attr_da(vi[2 * i0 + 1], vi[2 * i1 + 1], vi[2 * i2 + 1], db, &da[2], &da[3]);
        }
        float px[4];
        mip_fetch(img, (size_t)v, cu, cv, da, px);
        const float weight = px[3] * vz;
        acc[0] += px[0] * weight; acc[1] += px[1] * weight; acc[2] += px[2] * weight; acc[3] += weight;
    }
    reinterpret_cast<f32x4*>(accum)[t] = acc;
}

int check_tex(const char* who, int Bt, int H, int W, int C, int max_level) {
    MVE_CHECK(Bt > 0 && H > 0 && W > 0 && C > 0 && C <= TM_MAX_C, MVE_ERR_ARG, "%s: bad texture shape [%d,%d,%d,%d] (channels <= %d)", who, Bt, H, W, C, TM_MAX_C);
    MVE_CHECK(max_level >= 0 && max_level <= 30, MVE_ERR_ARG, "%s: bad max_level %d", who, max_level);
    for (int l = 0; l < max_level; ++l) {
        const int h = tm_dim(H, l), w = tm_dim(W, l);
        MVE_CHECK((h == 1 || h % 2 == 0) && (w == 1 || w % 2 == 0), MVE_ERR_ARG,
                  "%s: mip level %d of a %dx%d texture has an odd extent (nvdiffrast refuses it too: clamp max_mip_level)", who, l + 1, H, W);
    }
    return MVE_OK;
}

TexDesc make_desc(const float* tex0, const float* mips, int Bt, int H, int W, int C, int max_level, bool shared) {
    TexDesc t;
    t.tex0 = tex0; t.mips = mips; t.H = H; t.W = W; t.C = C; t.max_level = max_level;
    t.tex_stride = shared ? 0 : (size_t)H * W * C;
    t.mip_stride = shared ? 0 : (size_t)tm_mip_offset(H, W, max_level + 1) * C;
    (void)Bt;
    return t;
}

}  // namespace

extern "C" {

int mve_mip_levels(int H, int W) {                 // levels above level 0 of the full stack (down to 1 x 1)
    int l = 0;
    while ((tm_dim(H, l) | tm_dim(W, l)) > 1) ++l;
    return l;
}

size_t mve_mip_texels(int H, int W, int max_level) { return (size_t)tm_mip_offset(H, W, max_level + 1); }

int mve_rasterize_db(const float* d_pos, int B, int V, const int32_t* d_tri, int F, const float* d_rast, int H, int W, float* d_rast_db,
                     void* stream) {
    const size_t total = (size_t)B * H * W;
    if (total == 0) return MVE_OK;
    MVE_CHECK(d_pos && d_tri && d_rast && d_rast_db, MVE_ERR_ARG, "rasterize_db: null pointer");
    k_rasterize_db<<<mve_cdiv(total, NT), NT, 0, (hipStream_t)stream>>>(d_pos, V, d_tri, F, d_rast, B, H, W, d_rast_db);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_interpolate_da(const float* d_attr, int attr_batch, int V, int C, const float* d_rast, const float* d_rast_db, int B, int npix,
                       const int32_t* d_tri, int F, float* d_out, void* stream) {
    const size_t total = (size_t)B * npix;
    if (total == 0 || C == 0) return MVE_OK;
    MVE_CHECK(d_attr && d_rast && d_rast_db && d_tri && d_out, MVE_ERR_ARG, "interpolate_da: null pointer");
    MVE_CHECK(attr_batch == 1 || attr_batch == B, MVE_ERR_ARG, "interpolate_da: attribute batch %d vs %d images", attr_batch, B);
    k_interpolate_da<<<mve_cdiv(total, NT), NT, 0, (hipStream_t)stream>>>(d_attr, attr_batch == 1 ? 0 : (size_t)V * C, C, d_rast, d_rast_db,
                                                                         total, (size_t)npix, d_tri, F, d_out);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_mip_build(const float* d_tex0, int Bt, int H, int W, int C, int max_level, float* d_mips, void* stream) {
    if (int rc = check_tex("mip_build", Bt, H, W, C, max_level)) return rc;
    if (max_level == 0) return MVE_OK;
    MVE_CHECK(d_tex0 && d_mips, MVE_ERR_ARG, "mip_build: null pointer");
    const size_t ms = (size_t)tm_mip_offset(H, W, max_level + 1) * C;
    int l = 0;
    while (l < max_level) {
        const int h = tm_dim(H, l), w = tm_dim(W, l);
        const float* src = l == 0 ? d_tex0 : d_mips + (size_t)tm_mip_offset(H, W, l) * C;
        const size_t ss = l == 0 ? (size_t)H * W * C : ms;
        const int th = h % 32 == 0 ? 32 : (h <= 32 ? h : 0), tw = w % 32 == 0 ? 32 : (w <= 32 ? w : 0);
        if (th && tw) {                                  // a tile walk: up to five levels in one launch
            int nl = 0, eh = th, ew = tw;
            MipTileArgs a;
            while (nl < 5 && l + nl < max_level && (eh > 1 || ew > 1)) {
                a.off[nl] = tm_mip_offset(H, W, l + nl + 1);
                eh = eh > 1 ? eh >> 1 : 1; ew = ew > 1 ? ew >> 1 : 1;
                ++nl;
            }
            if (nl > 0) {
                const size_t lds = sizeof(float) * ((size_t)th * tw * C + (size_t)th * tw * C / 2 + C);
                k_mip_build_tile<<<dim3((h / th) * (w / tw), Bt), NT, lds, (hipStream_t)stream>>>(src, ss, d_mips, ms, h, w, C, th, tw, nl, a);
                MVE_LAUNCH_CHECK();
                l += nl;
                continue;
            }
        }
        const int h2 = tm_dim(H, l + 1), w2 = tm_dim(W, l + 1);
        k_mip_build<<<mve_cdiv((size_t)Bt * h2 * w2 * C, NT), NT, 0, (hipStream_t)stream>>>(
            src, ss, d_mips + (size_t)tm_mip_offset(H, W, l + 1) * C, ms, Bt, h, w, C);
        MVE_LAUNCH_CHECK();
        ++l;
    }
    return MVE_OK;
}

int mve_texture_mip(const float* d_tex0, const float* d_mips, int Bt, int H, int W, int C, int max_level, const float* d_uv,
                    const float* d_uv_da, const float* d_rast, int n, int h, int w, float* d_out, void* stream) {
    const size_t total = (size_t)n * h * w;
    if (total == 0) return MVE_OK;
    if (int rc = check_tex("texture_mip", Bt, H, W, C, max_level)) return rc;
    MVE_CHECK(d_tex0 && (d_mips || max_level == 0) && d_uv && d_uv_da && d_out, MVE_ERR_ARG, "texture_mip: null pointer");
    MVE_CHECK(Bt == 1 || Bt == n, MVE_ERR_ARG, "texture_mip: %d textures for %d images", Bt, n);
    const TexDesc t = make_desc(d_tex0, d_mips, Bt, H, W, C, max_level, Bt == 1);
    k_texture_mip<<<mve_cdiv(total, NT), NT, 0, (hipStream_t)stream>>>(t, d_uv, d_uv_da, d_rast, total, (size_t)h * w, d_out);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_texture_mip_backward(const float* d_g_out, int Bt, int H, int W, int C, int max_level, const float* d_uv, const float* d_uv_da,
                             const float* d_rast, int n, int h, int w, float* d_g_tex0, float* d_g_mips, void* stream) {
    if (int rc = check_tex("texture_mip_backward", Bt, H, W, C, max_level)) return rc;
    MVE_CHECK(d_g_out && d_uv && d_uv_da && d_g_tex0 && (d_g_mips || max_level == 0), MVE_ERR_ARG, "texture_mip_backward: null pointer");
    MVE_CHECK(Bt == 1 || Bt == n, MVE_ERR_ARG, "texture_mip_backward: %d textures for %d images", Bt, n);
    hipStream_t s = (hipStream_t)stream;
    const size_t mt = (size_t)tm_mip_offset(H, W, max_level + 1) * C;
    MVE_HIP(hipMemsetAsync(d_g_tex0, 0, sizeof(float) * (size_t)Bt * H * W * C, s));
    if (mt) MVE_HIP(hipMemsetAsync(d_g_mips, 0, sizeof(float) * (size_t)Bt * mt, s));
    TexDesc t = make_desc(nullptr, nullptr, Bt, H, W, C, max_level, Bt == 1);
    const size_t total = (size_t)n * h * w;
    if (total) {
        k_texture_mip_bwd<false><<<mve_cdiv(total, NT), NT, 0, s>>>(t, d_g_out, d_uv, d_uv_da, d_rast, total, (size_t)h * w, d_g_tex0, d_g_mips);
        MVE_LAUNCH_CHECK();
    }
    if (max_level > 0) {
        t.mip_stride = mt;           // the collapse walks every texture, shared or not
        launch_collapse<false>(t, Bt, d_g_mips, d_g_tex0, nullptr, s);
        MVE_LAUNCH_CHECK();
    }
    return MVE_OK;
}

size_t mve_visibility_mip_workspace_bytes(int n, int map_size, int max_level) {
    return sizeof(unsigned long long) * (size_t)n * ((size_t)map_size * map_size + (size_t)tm_mip_offset(map_size, map_size, max_level + 1));
}

// visibility_grad of get_cam_weights_uv / bake_multiview: d sum(dr.texture(ones [n,map,map,1], texc, uv_da=texc_db)) / d ones, per view
int mve_visibility_mip(const float* d_texc, const float* d_texc_da, const float* d_rast, int n, int h, int w, int map_size, int max_level,
                       void* d_workspace, size_t workspace_bytes, float* d_vis, void* stream) {
    if (n == 0) return MVE_OK;
    if (int rc = check_tex("visibility_mip", n, map_size, map_size, 1, max_level)) return rc;
    MVE_CHECK(d_texc && d_texc_da && d_rast && d_workspace && d_vis, MVE_ERR_ARG, "visibility_mip: null pointer");
    MVE_CHECK(workspace_bytes >= mve_visibility_mip_workspace_bytes(n, map_size, max_level), MVE_ERR_ARG, "visibility_mip: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    MVE_HIP(hipMemsetAsync(d_workspace, 0, mve_visibility_mip_workspace_bytes(n, map_size, max_level), s));
    unsigned long long* g0 = static_cast<unsigned long long*>(d_workspace);
    unsigned long long* gm = g0 + (size_t)n * map_size * map_size;
    TexDesc t = make_desc(nullptr, nullptr, n, map_size, map_size, 1, max_level, false);
    const size_t total = (size_t)n * h * w;
    k_texture_mip_bwd<true><<<mve_cdiv(total, NT), NT, 0, s>>>(t, nullptr, d_texc, d_texc_da, d_rast, total, (size_t)h * w, g0, gm);
    MVE_LAUNCH_CHECK();
    launch_collapse<true>(t, n, gm, g0, d_vis, s);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_bake_accumulate_mip(const float* d_tex_rast, const float* d_tex_rast_db, const int32_t* d_f, int F, const float* d_v_img, int V,
                            const float* d_img0, const float* d_img_mips, int h, int w, int max_level, const float* d_vis, int n,
                            int map_size, float* d_accum, void* stream) {
    if (n == 0 || map_size == 0) return MVE_OK;
    if (int rc = check_tex("bake_accumulate_mip", n, h, w, 4, max_level)) return rc;
    MVE_CHECK(d_tex_rast && d_tex_rast_db && d_f && d_v_img && d_img0 && (d_img_mips || max_level == 0) && d_vis && d_accum, MVE_ERR_ARG,
              "bake_accumulate_mip: null pointer");
    const TexDesc img = make_desc(d_img0, d_img_mips, n, h, w, 4, max_level, false);
    k_bake_accumulate_mip<<<mve_cdiv((size_t)map_size * map_size, NT), NT, 0, (hipStream_t)stream>>>(
        d_tex_rast, d_tex_rast_db, d_f, F, d_v_img, V, img, d_vis, n, map_size, d_accum);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
