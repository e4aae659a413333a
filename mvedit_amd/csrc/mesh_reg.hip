// Mesh regularisers of the mesh-optimisation loop (gfx950; HBM / latency bound): laplacian_smooth_loss and normal_consistency
// (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:20-101, called per iteration at lib/pipelines/mvedit_3d_pipeline.py:775-776)
// with their gradients, without the reference's two device-wide torch.unique sorts per iteration: the half-edges are scattered into
// per-vertex buckets (count -> scan -> fill), one thread sorts its vertex's bucket (a dozen 64-bit entries) and reads the distinct
// neighbours (Laplacian row) and the edge -> face pairs off it.  Arithmetic in mesh_reg_core.h (host/device; CPU-tested against the
// reference's own functions).  The atomics of the fill pass only decide the order inside a bucket BEFORE it is sorted; the loss sums
// are fixed-order trees: the forward is bitwise reproducible.  The face-normal gradient is a float-atomic scatter (3 adds per face).
#include "common.h"
#include "reduce.h"
#include "scan.h"

#include "mesh_reg_core.h"

namespace {

constexpr int NT = 256;

struct Ws {
    int *cnt, *base, *fill, *tile, *total, *ne_part;   // total[0] = 6F (scan total), total[1] = E
    uint64_t* bucket;
    float *u, *part;                                   // part[2][nbv]
    unsigned nbv;
};

size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

size_t ws_bytes_of(size_t V, size_t F) {
    const size_t nbv = mve_cdiv(V, NT), ntiles = (V + SCAN_TILE - 1) / SCAN_TILE;
    return align8((3 * V + ntiles + 2 + nbv) * sizeof(int)) + 6 * F * sizeof(uint64_t) + (3 * V + 2 * nbv) * sizeof(float);
}

Ws carve(void* ws, size_t V, size_t F) {
    Ws w;
    w.nbv = (unsigned)mve_cdiv(V, NT);
    const size_t ntiles = (V + SCAN_TILE - 1) / SCAN_TILE;
    int* ip = static_cast<int*>(ws);
    w.cnt = ip; w.base = ip + V; w.fill = ip + 2 * V; w.tile = ip + 3 * V; w.total = w.tile + ntiles; w.ne_part = w.total + 2;
    unsigned char* b = static_cast<unsigned char*>(ws) + align8((3 * V + ntiles + 2 + w.nbv) * sizeof(int));
    w.bucket = reinterpret_cast<uint64_t*>(b);
    w.u = reinterpret_cast<float*>(b + 6 * F * sizeof(uint64_t));
    w.part = w.u + 3 * V;
    return w;
}

__global__ __launch_bounds__(NT) void k_mr_count(const int32_t* __restrict__ faces, int F, int* __restrict__ cnt) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= F) return;
    for (int k = 0; k < 3; ++k) atomicAdd(cnt + faces[3 * t + k], 2);
}

__global__ __launch_bounds__(NT) void k_mr_fill(const int32_t* __restrict__ faces, int F, const int* __restrict__ base, int* __restrict__ fill,
                                                uint64_t* __restrict__ bucket) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= F) return;
    const int v[3] = {faces[3 * t], faces[3 * t + 1], faces[3 * t + 2]};
    for (int k = 0; k < 3; ++k) {
        const int a = v[k], b = v[(k + 1) % 3], side = a > b ? 1 : 0;
        bucket[base[a] + atomicAdd(fill + a, 1)] = mr_pack(b, t, side);
        bucket[base[b] + atomicAdd(fill + b, 1)] = mr_pack(a, t, side);
    }
}

__global__ __launch_bounds__(NT) void k_mr_vertex_fwd(const float* __restrict__ verts, int V, const float* __restrict__ face_normals, Ws w) {
    __shared__ float sh[NT];
    const int i = blockIdx.x * NT + threadIdx.x;
    float lap = 0.f, nc = 0.f;
    int ne = 0;
    if (i < V) {
        uint64_t* b = w.bucket + w.base[i];
        mr_sort(b, w.cnt[i]);
        lap = mr_vertex_fwd(i, b, w.cnt[i], verts, face_normals, w.u + 3 * i, &nc, &ne);
    }
    const float s0 = mve_block_sum<NT>(lap, sh), s1 = mve_block_sum<NT>(nc, sh), s2 = mve_block_sum<NT>((float)ne, sh);   // ne <= a few thousand per block: exact in f32
    if (threadIdx.x == 0) { w.part[blockIdx.x] = s0; w.part[w.nbv + blockIdx.x] = s1; w.ne_part[blockIdx.x] = (int)s2; }
}

__global__ __launch_bounds__(NT) void k_mr_reduce(Ws w, int V, float* __restrict__ losses) {
    __shared__ float sh[NT];
    __shared__ int shi[NT];
    float a = 0.f, b = 0.f;
    int e = 0;
    for (unsigned c = threadIdx.x; c < w.nbv; c += NT) { a += w.part[c]; b += w.part[w.nbv + c]; e += w.ne_part[c]; }
    const float lap = mve_block_sum<NT>(a, sh), nc = mve_block_sum<NT>(b, sh);
    shi[threadIdx.x] = e;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) shi[threadIdx.x] += shi[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int E = shi[0];
        w.total[1] = E;
        losses[0] = lap / (float)V;
        losses[1] = E > 0 ? nc / (float)E : 0.f;
    }
}

__global__ __launch_bounds__(NT) void k_mr_vertex_bwd(int V, const float* __restrict__ face_normals, Ws w, const float* __restrict__ d_gl,
                                                      float* __restrict__ g_verts, float* __restrict__ g_fn) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= V) return;
    const float gl_lap = d_gl ? d_gl[0] : 1.0f, gl_nc = d_gl ? d_gl[1] : 1.0f;
    const int E = w.total[1];
    const uint64_t* b = w.bucket + w.base[i];
    mr_vertex_bwd_lap(i, b, w.cnt[i], w.u, gl_lap / (float)V, g_verts);
    if (E > 0) mr_vertex_bwd_nc(i, b, w.cnt[i], face_normals, gl_nc / (float)E, g_fn);
}

// ---- Mesh.auto_normal -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_mn_face_fwd(const float* __restrict__ verts, const int32_t* __restrict__ faces, int F,
                                                    float* __restrict__ face_normals, float* __restrict__ vn_sum) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t < F) mr_normals_face_fwd(verts, faces, t, face_normals, vn_sum);
}

__global__ __launch_bounds__(NT) void k_mn_vertex_fwd(int V, const float* __restrict__ vn_sum, float* __restrict__ vn) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= V) return;
    const float l = sqrtf(mr_dot3(vn_sum + 3 * i, vn_sum + 3 * i)), inv = 1.0f / fmaxf(l, 1e-12f);
    for (int k = 0; k < 3; ++k) vn[3 * i + k] = vn_sum[3 * i + k] * inv;
}

__global__ __launch_bounds__(NT) void k_mn_vertex_bwd(int V, const float* __restrict__ vn_sum, const float* __restrict__ g_vn, float* __restrict__ g_sum) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= V) return;
    const float zero[3] = {0.f, 0.f, 0.f};
    mr_normalize_bwd(vn_sum + 3 * i, g_vn ? g_vn + 3 * i : zero, g_sum + 3 * i);
}

__global__ __launch_bounds__(NT) void k_mn_face_bwd(const float* __restrict__ verts, const int32_t* __restrict__ faces, int F,
                                                    const float* __restrict__ g_fn_ext, const float* __restrict__ g_sum, float* __restrict__ g_verts) {
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t < F) mr_normals_face_bwd(verts, faces, t, g_fn_ext, g_sum, g_verts);
}

int check(const char* who, const float* verts, int V, const int32_t* faces, int F, const float* fn, const void* ws, size_t ws_bytes) {
    MVE_CHECK(verts && faces && fn && ws, MVE_ERR_ARG, "%s: null pointer", who);
    MVE_CHECK(V > 0 && F > 0 && (long long)F * 6 < (1ll << 31), MVE_ERR_ARG, "%s: bad mesh size V=%d F=%d", who, V, F);
    MVE_CHECK(ws_bytes >= ws_bytes_of(V, F), MVE_ERR_ARG, "%s: workspace %zu < %zu bytes", who, ws_bytes, ws_bytes_of(V, F));
    return MVE_OK;
}

}  // namespace

extern "C" {

size_t mve_mesh_reg_workspace_bytes(int V, int F) { return (V > 0 && F > 0) ? ws_bytes_of(V, F) : 0; }

int mve_mesh_reg_forward(const float* d_verts, int V, const int32_t* d_faces, int F, const float* d_face_normals, void* d_ws, size_t ws_bytes,
                         float* d_losses, void* stream) {
    if (int rc = check("mesh_reg_forward", d_verts, V, d_faces, F, d_face_normals, d_ws, ws_bytes)) return rc;
    MVE_CHECK(d_losses, MVE_ERR_ARG, "mesh_reg_forward: null output");
    hipStream_t s = (hipStream_t)stream;
    const Ws w = carve(d_ws, V, F);
    MVE_HIP(hipMemsetAsync(w.cnt, 0, sizeof(int) * (size_t)V, s));
    MVE_HIP(hipMemsetAsync(w.fill, 0, sizeof(int) * (size_t)V, s));
    const unsigned nbf = mve_cdiv(F, NT);
    k_mr_count<<<nbf, NT, 0, s>>>(d_faces, F, w.cnt);
    MVE_LAUNCH_CHECK();
    if (int rc = exclusive_scan<int>(w.cnt, w.base, (size_t)V, w.tile, w.total, s)) { mve_set_error("mesh_reg_forward: scan launch failed"); return rc; }
    k_mr_fill<<<nbf, NT, 0, s>>>(d_faces, F, w.base, w.fill, w.bucket);
    MVE_LAUNCH_CHECK();
    k_mr_vertex_fwd<<<w.nbv, NT, 0, s>>>(d_verts, V, d_face_normals, w);
    MVE_LAUNCH_CHECK();
    k_mr_reduce<<<1, NT, 0, s>>>(w, V, d_losses);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_mesh_reg_backward(const float* d_verts, int V, const int32_t* d_faces, int F, const float* d_face_normals, void* d_ws, size_t ws_bytes,
                          const float* d_g_losses, float* d_g_verts, float* d_g_face_normals, void* stream) {
    if (int rc = check("mesh_reg_backward", d_verts, V, d_faces, F, d_face_normals, d_ws, ws_bytes)) return rc;
    MVE_CHECK(d_g_verts && d_g_face_normals, MVE_ERR_ARG, "mesh_reg_backward: null output");
    hipStream_t s = (hipStream_t)stream;
    const Ws w = carve(d_ws, V, F);
    MVE_HIP(hipMemsetAsync(d_g_face_normals, 0, sizeof(float) * 3 * (size_t)F, s));
    k_mr_vertex_bwd<<<w.nbv, NT, 0, s>>>(V, d_face_normals, w, d_g_losses, d_g_verts, d_g_face_normals);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_mesh_normals_forward(const float* d_verts, int V, const int32_t* d_faces, int F, float* d_face_normals, float* d_vn_sum, float* d_vn,
                             void* stream) {
    MVE_CHECK(d_verts && d_faces && d_face_normals && d_vn_sum && d_vn, MVE_ERR_ARG, "mesh_normals_forward: null pointer");
    MVE_CHECK(V > 0 && F > 0, MVE_ERR_ARG, "mesh_normals_forward: bad mesh size V=%d F=%d", V, F);
    hipStream_t s = (hipStream_t)stream;
    MVE_HIP(hipMemsetAsync(d_vn_sum, 0, sizeof(float) * 3 * (size_t)V, s));
    k_mn_face_fwd<<<mve_cdiv(F, NT), NT, 0, s>>>(d_verts, d_faces, F, d_face_normals, d_vn_sum);
    MVE_LAUNCH_CHECK();
    k_mn_vertex_fwd<<<mve_cdiv(V, NT), NT, 0, s>>>(V, d_vn_sum, d_vn);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_mesh_normals_backward(const float* d_verts, int V, const int32_t* d_faces, int F, const float* d_vn_sum, const float* d_g_vn,
                              const float* d_g_face_normals, float* d_scratch, float* d_g_verts, void* stream) {
    MVE_CHECK(d_verts && d_faces && d_vn_sum && d_scratch && d_g_verts, MVE_ERR_ARG, "mesh_normals_backward: null pointer");
    MVE_CHECK(V > 0 && F > 0, MVE_ERR_ARG, "mesh_normals_backward: bad mesh size V=%d F=%d", V, F);
    hipStream_t s = (hipStream_t)stream;
    MVE_HIP(hipMemsetAsync(d_g_verts, 0, sizeof(float) * 3 * (size_t)V, s));
    k_mn_vertex_bwd<<<mve_cdiv(V, NT), NT, 0, s>>>(V, d_vn_sum, d_g_vn, d_scratch);
    MVE_LAUNCH_CHECK();
    k_mn_face_bwd<<<mve_cdiv(F, NT), NT, 0, s>>>(d_verts, d_faces, F, d_g_face_normals, d_scratch, d_g_verts);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
