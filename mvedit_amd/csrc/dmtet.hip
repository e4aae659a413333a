// Marching tetrahedra (DMTet.__call__, lib/models/decoders/mesh_renderer/base_mesh_renderer.py:140-188) without the sort.
//
// The reference extracts the iso-surface with torch.unique(dim=0) over the (sorted) edges of all surface-crossing tetrahedra: a
// lexicographic sort of ~6 * valid_tets int64 pairs per mesh-optimisation iteration.  The output ORDER matters for parity:
// vertex k is the k-th crossing edge (a < b) in lexicographic order; faces list all one-triangle tets first, then the
// two-triangle tets, each group in tet order.
//
// MI355X design: the rank of edge (a, b) is  sum_{a' < a} deg(a') + rank of b among a's crossing partners,  and a vertex of a tet
// grid has at most a few dozen incident edge instances.  So: count crossing edge instances per lower endpoint (atomics on
// per-vertex counters), exclusive-scan the counters, bucket the upper endpoints, sort + de-duplicate every bucket with one
// thread per vertex (buckets are tiny), scan the unique counts -> vertex numbering identical to the sorted order, bit for bit,
// with HBM traffic linear in the number of tets and no global sort.  Face offsets come from one more scan over the tets.
// Determinism: the atomics only decide the order inside a bucket BEFORE it is sorted.
//
// Two-call protocol (the output sizes are data dependent, like march_rays_train): mve_dmtet_count -> host reads
// (n_verts, n_faces) -> mve_dmtet_write into exactly sized buffers; the workspace carries the state between the calls.
// Compiled with -ffp-contract=off: the interpolation must round like the reference's separate mul / add / div.
#include "common.h"
#include "scan.h"

namespace {

constexpr int DB = 256;

__constant__ signed char c_tri_table[16][6] = {
    {-1, -1, -1, -1, -1, -1}, {1, 0, 2, -1, -1, -1}, {4, 0, 3, -1, -1, -1}, {1, 4, 2, 1, 3, 4}, {3, 1, 5, -1, -1, -1}, {2, 3, 0, 2, 5, 3},
    {1, 4, 0, 1, 5, 4},       {4, 2, 5, -1, -1, -1}, {4, 5, 2, -1, -1, -1}, {4, 1, 0, 4, 5, 1}, {3, 2, 0, 3, 5, 2},    {1, 3, 5, -1, -1, -1},
    {4, 1, 2, 4, 3, 1},       {3, 0, 4, -1, -1, -1}, {2, 0, 1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}};
__constant__ signed char c_num_tri[16] = {0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0};
__constant__ signed char c_edge_a[6] = {0, 0, 0, 1, 1, 2};
__constant__ signed char c_edge_b[6] = {1, 2, 3, 2, 3, 3};

__device__ __forceinline__ int tet_code(const float* __restrict__ sdf, const int32_t* __restrict__ tet) {
    int code = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) code |= (sdf[tet[k]] > 0.0f ? 1 : 0) << k;
    return code;
}

// pass 1/2: visit the crossing edges of the valid tets; MODE 0 counts per lower endpoint, MODE 1 buckets the upper endpoints
template <int MODE>
__global__ __launch_bounds__(DB) void k_tet_edges(const float* __restrict__ sdf, const int32_t* __restrict__ tets, size_t Nt,
                                                  int* __restrict__ cnt, const int* __restrict__ base, int* __restrict__ bucket,
                                                  unsigned long long* __restrict__ tri_flags) {
    const size_t t = (size_t)blockIdx.x * DB + threadIdx.x;
    if (t >= Nt) return;
    int v[4];
    *reinterpret_cast<int4*>(v) = reinterpret_cast<const int4*>(tets)[t];
    const int code = tet_code(sdf, v);
    if (MODE == 0) {
        const int nt = c_num_tri[code];
        tri_flags[t] = nt == 1 ? 1ull : (nt == 2 ? (1ull << 32) : 0ull);
    }
    if (code == 0 || code == 15) return;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int ia = c_edge_a[e], ib = c_edge_b[e];
        if (((code >> ia) & 1) == ((code >> ib) & 1)) continue;
        const int a = v[ia] < v[ib] ? v[ia] : v[ib], b = v[ia] < v[ib] ? v[ib] : v[ia];
        const int slot = atomicAdd(cnt + a, 1);
        if (MODE == 1) bucket[base[a] + slot] = b;
    }
}

// one thread per vertex: sort its bucket, drop duplicates, remember how many distinct partners remain
__global__ __launch_bounds__(DB) void k_bucket_unique(const int* __restrict__ base, const int* __restrict__ cnt, int Nv, int* __restrict__ bucket,
                                                      int* __restrict__ uniq) {
    const int a = blockIdx.x * DB + threadIdx.x;
    if (a >= Nv) return;
    const int n = cnt[a];
    int* b = bucket + base[a];
    for (int i = 1; i < n; ++i) {
        const int key = b[i];
        int j = i - 1;
        while (j >= 0 && b[j] > key) { b[j + 1] = b[j]; --j; }
        b[j + 1] = key;
    }
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (i == 0 || b[i] != b[m - 1]) b[m++] = b[i];
    uniq[a] = m;
}

__global__ __launch_bounds__(DB) void k_interp_verts(const float* __restrict__ pos, const float* __restrict__ sdf, const int* __restrict__ base,
                                                     const int* __restrict__ uniq, const int* __restrict__ ubase, const int* __restrict__ bucket,
                                                     int Nv, float* __restrict__ verts, int32_t* __restrict__ edges) {
    const int a = blockIdx.x * DB + threadIdx.x;
    if (a >= Nv) return;
    const int m = uniq[a];
    const int* b = bucket + base[a];
    const float sa = sdf[a];
    for (int j = 0; j < m; ++j) {
        const int vb = b[j];
        // edges_to_interp_sdf = [s_a, -s_b]; denominator = s_a - s_b; weights = flip / denominator  (:170-176)
        const float nb = sdf[vb] * -1.0f;
        const float den = sa + nb;
        const float wa = nb / den, wb = sa / den;
        if (verts) {
            float* o = verts + 3ll * (ubase[a] + j);
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = pos[3ll * a + k] * wa + pos[3ll * vb + k] * wb;
        }
        if (edges) { edges[2ll * (ubase[a] + j)] = a; edges[2ll * (ubase[a] + j) + 1] = vb; }
    }
}

__global__ __launch_bounds__(DB) void k_emit_faces(const float* __restrict__ sdf, const int32_t* __restrict__ tets, size_t Nt,
                                                   const unsigned long long* __restrict__ tri_off, unsigned long long tri_total_unused,
                                                   const unsigned long long* __restrict__ tri_total, const int* __restrict__ base,
                                                   const int* __restrict__ uniq, const int* __restrict__ ubase, const int* __restrict__ bucket,
                                                   int32_t* __restrict__ faces) {
    const size_t t = (size_t)blockIdx.x * DB + threadIdx.x;
    if (t >= Nt) return;
    int v[4];
    *reinterpret_cast<int4*>(v) = reinterpret_cast<const int4*>(tets)[t];
    const int code = tet_code(sdf, v);
    const int nt = c_num_tri[code];
    if (nt == 0) return;
    const unsigned long long off = tri_off[t];
    const unsigned n1 = (unsigned)(*tri_total & 0xffffffffull);
    const size_t row = nt == 1 ? (size_t)(off & 0xffffffffull) : (size_t)n1 + 2 * (size_t)(off >> 32);
    for (int k = 0; k < 3 * nt; ++k) {
        const int e = c_tri_table[code][k];
        const int ia = c_edge_a[e], ib = c_edge_b[e];
        const int a = v[ia] < v[ib] ? v[ia] : v[ib], b = v[ia] < v[ib] ? v[ib] : v[ia];
        const int* lst = bucket + base[a];
        int lo = 0, hi = uniq[a];                           // lower_bound: the table only references crossing edges
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (lst[mid] < b) lo = mid + 1; else hi = mid; }
        faces[3 * row + k] = ubase[a] + lo;
    }
}

// d verts -> d pos, d sdf.  v = pa * wa + pb * wb, wa = -sb / (sa - sb), wb = sa / (sa - sb)  (base_mesh_renderer.py:170-176), so
//   d pa = wa g,  d pb = wb g,  d sa = sb (g . (pa - pb)) / (sa - sb)^2,  d sb = -sa (g . (pa - pb)) / (sa - sb)^2.
__global__ __launch_bounds__(DB) void k_dmtet_backward(const float* __restrict__ pos, const float* __restrict__ sdf, const int32_t* __restrict__ edges,
                                                       int n_out, const float* __restrict__ g_verts, float* __restrict__ g_pos,
                                                       float* __restrict__ g_sdf) {
    const int k = blockIdx.x * DB + threadIdx.x;
    if (k >= n_out) return;
    const int a = edges[2 * k], b = edges[2 * k + 1];
    const float sa = sdf[a], sb = sdf[b];
    const float den = sa - sb, wa = -sb / den, wb = sa / den;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = g_verts[3ll * k + c];
        if (g_pos) { atomicAdd(g_pos + 3ll * a + c, wa * g); atomicAdd(g_pos + 3ll * b + c, wb * g); }
        dot += g * (pos[3ll * a + c] - pos[3ll * b + c]);
    }
    if (g_sdf) {
        const float r = dot / (den * den);
        atomicAdd(g_sdf + a, sb * r);
        atomicAdd(g_sdf + b, -sa * r);
    }
}

struct Layout {
    int *cnt, *base, *uniq, *ubase, *bucket, *tile_i32, *counts_i32;
    unsigned long long *flags, *off, *tile_u64, *total_u64;
    size_t bytes;
};
Layout layout(void* ws, size_t Nv, size_t Nt) {
    Layout L;
    unsigned char* p = (unsigned char*)ws;
    auto take = [&](size_t bytes) { unsigned char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    const size_t tiles_v = (Nv + SCAN_TILE - 1) / SCAN_TILE + 1, tiles_t = (Nt + SCAN_TILE - 1) / SCAN_TILE + 1;
    L.flags = (unsigned long long*)take(Nt * 8);
    L.off = (unsigned long long*)take(Nt * 8);
    L.tile_u64 = (unsigned long long*)take(tiles_t * 8);
    L.total_u64 = (unsigned long long*)take(8);
    L.cnt = (int*)take(Nv * 4);
    L.base = (int*)take(Nv * 4);
    L.uniq = (int*)take(Nv * 4);
    L.ubase = (int*)take(Nv * 4);
    L.tile_i32 = (int*)take(tiles_v * 4);
    L.counts_i32 = (int*)take(16);
    L.bucket = (int*)take(Nt * 6 * 4);
    L.bytes = (size_t)(p - (unsigned char*)ws);
    return L;
}

__global__ void k_dmtet_counts(const int* __restrict__ n_verts, const unsigned long long* __restrict__ tri_total, int32_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = *n_verts;
        out[1] = (int32_t)((*tri_total & 0xffffffffull) + 2 * (*tri_total >> 32));
    }
}

}  // namespace

extern "C" {

size_t mve_dmtet_workspace_bytes(size_t n_verts, size_t n_tets) { return layout(nullptr, n_verts, n_tets).bytes + 256; }

int mve_dmtet_count(const float* d_sdf, const int32_t* d_tets, size_t n_verts, size_t n_tets, int32_t* d_counts /* [2] */, void* d_workspace,
                    size_t workspace_bytes, void* stream) {
    MVE_CHECK(d_counts, MVE_ERR_ARG, "dmtet_count: null counts");
    hipStream_t s = (hipStream_t)stream;
    if (n_tets == 0 || n_verts == 0) { MVE_HIP(hipMemsetAsync(d_counts, 0, 8, s)); return MVE_OK; }
    MVE_CHECK(d_sdf && d_tets && d_workspace, MVE_ERR_ARG, "dmtet_count: null pointer");
    MVE_CHECK(workspace_bytes >= mve_dmtet_workspace_bytes(n_verts, n_tets), MVE_ERR_NOMEM, "dmtet_count: workspace too small");
    MVE_CHECK(n_verts < (1ull << 31) && n_tets < (1ull << 31), MVE_ERR_ARG, "dmtet_count: sizes must fit in int32");
    Layout L = layout(d_workspace, n_verts, n_tets);
    const unsigned gt = (unsigned)((n_tets + DB - 1) / DB), gv = (unsigned)((n_verts + DB - 1) / DB);
    MVE_HIP(hipMemsetAsync(L.cnt, 0, n_verts * 4, s));
    k_tet_edges<0><<<gt, DB, 0, s>>>(d_sdf, d_tets, n_tets, L.cnt, nullptr, nullptr, L.flags);
    MVE_LAUNCH_CHECK();
    int rc = exclusive_scan<int>(L.cnt, L.base, n_verts, L.tile_i32, L.counts_i32 + 1, s);
    if (rc) return rc;
    MVE_HIP(hipMemsetAsync(L.cnt, 0, n_verts * 4, s));
    k_tet_edges<1><<<gt, DB, 0, s>>>(d_sdf, d_tets, n_tets, L.cnt, L.base, L.bucket, nullptr);
    MVE_LAUNCH_CHECK();
    k_bucket_unique<<<gv, DB, 0, s>>>(L.base, L.cnt, (int)n_verts, L.bucket, L.uniq);
    MVE_LAUNCH_CHECK();
    rc = exclusive_scan<int>(L.uniq, L.ubase, n_verts, L.tile_i32, L.counts_i32, s);
    if (rc) return rc;
    rc = exclusive_scan<unsigned long long>(L.flags, L.off, n_tets, L.tile_u64, L.total_u64, s);
    if (rc) return rc;
    k_dmtet_counts<<<1, 64, 0, s>>>(L.counts_i32, L.total_u64, d_counts);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_dmtet_write(const float* d_pos, const float* d_sdf, const int32_t* d_tets, size_t n_verts, size_t n_tets, float* d_out_verts,
                    int32_t* d_out_faces, int32_t* d_out_edges, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (n_tets == 0 || n_verts == 0) return MVE_OK;
    MVE_CHECK(d_pos && d_sdf && d_tets && d_workspace, MVE_ERR_ARG, "dmtet_write: null pointer");
    MVE_CHECK(workspace_bytes >= mve_dmtet_workspace_bytes(n_verts, n_tets), MVE_ERR_NOMEM, "dmtet_write: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Layout L = layout(d_workspace, n_verts, n_tets);
    const unsigned gt = (unsigned)((n_tets + DB - 1) / DB), gv = (unsigned)((n_verts + DB - 1) / DB);
    if (d_out_verts || d_out_edges) {
        k_interp_verts<<<gv, DB, 0, s>>>(d_pos, d_sdf, L.base, L.uniq, L.ubase, L.bucket, (int)n_verts, d_out_verts, d_out_edges);
        MVE_LAUNCH_CHECK();
    }
    if (d_out_faces) {
        k_emit_faces<<<gt, DB, 0, s>>>(d_sdf, d_tets, n_tets, L.off, 0ull, L.total_u64, L.base, L.uniq, L.ubase, L.bucket, d_out_faces);
        MVE_LAUNCH_CHECK();
    }
    return MVE_OK;
}

int mve_dmtet_backward(const float* d_pos, const float* d_sdf, const int32_t* d_edges, size_t n_out_verts, const float* d_grad_verts,
                       float* d_grad_pos, float* d_grad_sdf, void* stream) {
    if (n_out_verts == 0) return MVE_OK;
    MVE_CHECK(d_pos && d_sdf && d_edges && d_grad_verts && (d_grad_pos || d_grad_sdf), MVE_ERR_ARG, "dmtet_backward: null pointer");
    k_dmtet_backward<<<(unsigned)((n_out_verts + DB - 1) / DB), DB, 0, (hipStream_t)stream>>>(d_pos, d_sdf, d_edges, (int)n_out_verts, d_grad_verts,
                                                                                               d_grad_pos, d_grad_sdf);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // extern "C"
