// Per-vertex arithmetic of the two mesh regularisers of the mesh-optimisation loop (mesh_reg.hip), written so that the same source also
// compiles for the host (oracle/devcore_host.cpp): the CPU tests run that build against the reference's own functions executed.
// Reference: lib/models/decoders/mesh_renderer/base_mesh_renderer.py -- compute_edge_to_face_mapping (:20-52), normal_consistency (:55-68),
// laplacian_uniform (:71-91), laplacian_smooth_loss (:94-101); called once per iteration at lib/pipelines/mvedit_3d_pipeline.py:775-776.
//
// The reference finds the unique edges with two torch.unique calls (device-wide sorts) per iteration.  Here every vertex owns a bucket of
// its incident half-edges (filled by mesh_reg.hip: count, scan, scatter); one thread sorts its bucket (a dozen entries) and reads off
//   * the distinct neighbours            -> row i of the uniform Laplacian  L = D - A  (laplacian_uniform)
//   * the edges {i, j > i} with the face on each side -> one row of tris_per_edge (compute_edge_to_face_mapping)
// Bucket entry = partner << 32 | face << 1 | side, side = 0 when the face lists the edge as (min, max), 1 when as (max, min)
// (`order` in the reference).  A missing side keeps the reference's default, face 0; when several faces claim one side of a non-manifold
// edge the reference's scatter keeps an arbitrary one, this code the highest-numbered.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MVE_MR_FN __device__ __forceinline__
#else
#define MVE_MR_FN static inline
#endif

MVE_MR_FN uint64_t mr_pack(int partner, int face, int side) { return ((uint64_t)(uint32_t)partner << 32) | ((uint64_t)(uint32_t)face << 1) | (uint64_t)side; }
MVE_MR_FN int mr_partner(uint64_t e) { return (int)(e >> 32); }
MVE_MR_FN int mr_face(uint64_t e) { return (int)((e & 0xffffffffull) >> 1); }
MVE_MR_FN int mr_side(uint64_t e) { return (int)(e & 1ull); }

MVE_MR_FN void mr_sort(uint64_t* b, int n) {
    for (int i = 1; i < n; ++i) {
        const uint64_t k = b[i];
        int j = i - 1;
        while (j >= 0 && b[j] > k) { b[j + 1] = b[j]; --j; }
        b[j + 1] = k;
    }
}

// the two faces of the edge whose entries are b[s..e) (same partner), sides 0 / 1
MVE_MR_FN void mr_edge_faces(const uint64_t* b, int s, int e, int* t0, int* t1) {
    *t0 = 0; *t1 = 0;
    for (int k = s; k < e; ++k) {
        if (mr_side(b[k])) *t1 = mr_face(b[k]);
        else *t0 = mr_face(b[k]);
    }
}

MVE_MR_FN float mr_dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// forward of vertex i over its SORTED bucket b[0..n): u[3] = (L v)_i, returns |u|; *nc_sum / *n_edges: sum of |1 - clamp(n0 . n1, -1, 1)|
// over the edges {i, j > i} and their number
MVE_MR_FN float mr_vertex_fwd(int i, const uint64_t* b, int n, const float* verts, const float* face_normals, float* u, float* nc_sum, int* n_edges) {
    float acc[3] = {0.f, 0.f, 0.f}, nc = 0.f;
    int deg = 0, ne = 0, s = 0;
    while (s < n) {
        const int j = mr_partner(b[s]);
        int e = s + 1;
        while (e < n && mr_partner(b[e]) == j) ++e;
        ++deg;
        for (int k = 0; k < 3; ++k) acc[k] += verts[3 * j + k];
        if (j > i) {
            int t0, t1;
            mr_edge_faces(b, s, e, &t0, &t1);
            const float d = mr_dot3(face_normals + 3 * t0, face_normals + 3 * t1);
            nc += fabsf(1.0f - fminf(fmaxf(d, -1.0f), 1.0f));
            ++ne;
        }
        s = e;
    }
    for (int k = 0; k < 3; ++k) u[k] = (float)deg * verts[3 * i + k] - acc[k];
    *nc_sum = nc; *n_edges = ne;
    return sqrtf(mr_dot3(u, u));
}

// d mean_i |u_i| / d v_i = (L^T uhat)_i / V with uhat = u / |u| (0 where u = 0, torch's norm backward); L is symmetric
MVE_MR_FN void mr_vertex_bwd_lap(int i, const uint64_t* b, int n, const float* u_all, float coef, float* g_v) {
    float acc[3] = {0.f, 0.f, 0.f}, hi[3];
    int deg = 0, s = 0;
    {
        const float l = sqrtf(mr_dot3(u_all + 3 * i, u_all + 3 * i)), inv = l > 0.f ? 1.0f / l : 0.f;
        for (int k = 0; k < 3; ++k) hi[k] = u_all[3 * i + k] * inv;
    }
    while (s < n) {
        const int j = mr_partner(b[s]);
        int e = s + 1;
        while (e < n && mr_partner(b[e]) == j) ++e;
        ++deg;
        const float l = sqrtf(mr_dot3(u_all + 3 * j, u_all + 3 * j)), inv = l > 0.f ? 1.0f / l : 0.f;
        for (int k = 0; k < 3; ++k) acc[k] += u_all[3 * j + k] * inv;
        s = e;
    }
    for (int k = 0; k < 3; ++k) g_v[3 * i + k] = coef * ((float)deg * hi[k] - acc[k]);
}

#if defined(__HIPCC__)
MVE_MR_FN void mr_add(float* p, float v) { atomicAdd(p, v); }
#else
MVE_MR_FN void mr_add(float* p, float v) { *p += v; }
#endif

// d mean_e |1 - clamp(n0 . n1)| / d face normals for the edges owned by vertex i: -coef * n1 into t0, -coef * n0 into t1 where the clamp and
// the abs pass (|dot| <= 1 and the term is positive); g_fn must be zero-initialised
MVE_MR_FN void mr_vertex_bwd_nc(int i, const uint64_t* b, int n, const float* face_normals, float coef, float* g_fn) {
    int s = 0;
    while (s < n) {
        const int j = mr_partner(b[s]);
        int e = s + 1;
        while (e < n && mr_partner(b[e]) == j) ++e;
        if (j > i) {
            int t0, t1;
            mr_edge_faces(b, s, e, &t0, &t1);
            const float* n0 = face_normals + 3 * t0;
            const float* n1 = face_normals + 3 * t1;
            const float d = mr_dot3(n0, n1);
            if (d >= -1.0f && d <= 1.0f && 1.0f - d > 0.f)
                for (int k = 0; k < 3; ++k) { mr_add(g_fn + 3 * t0 + k, -coef * n1[k]); mr_add(g_fn + 3 * t1 + k, -coef * n0[k]); }
        }
        s = e;
    }
}

// ---- Mesh.auto_normal (lib/models/decoders/mesh_renderer/mesh_utils.py:359-382, seamless=False) ------------------------------------------
// face normal = normalize(cross(v1 - v0, v2 - v0)) (F.normalize: / max(|.|, 1e-12)), splatted onto the three vertices (scatter_add -> float
// atomics, as in the reference), vertex normal = normalize(sum).
MVE_MR_FN void mr_face_normal(const float* verts, const int32_t* face, float* n, float* len) {
    const float* a = verts + 3 * face[0];
    const float* b = verts + 3 * face[1];
    const float* c = verts + 3 * face[2];
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const float cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const float l = sqrtf(mr_dot3(cr, cr)), inv = 1.0f / fmaxf(l, 1e-12f);
    *len = l;
    for (int k = 0; k < 3; ++k) n[k] = cr[k] * inv;
}

// face pass of the forward: face_normals[t], vn_sum[verts of t] += face normal
MVE_MR_FN void mr_normals_face_fwd(const float* verts, const int32_t* faces, int t, float* face_normals, float* vn_sum) {
    float n[3], l;
    mr_face_normal(verts, faces + 3 * t, n, &l);
    for (int k = 0; k < 3; ++k) face_normals[3 * t + k] = n[k];
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 3; ++k) mr_add(vn_sum + 3 * faces[3 * t + c] + k, n[k]);
}

// gradient of v / max(|v|, 1e-12) w.r.t. v given the incoming gradient g
MVE_MR_FN void mr_normalize_bwd(const float* v, const float* g, float* out) {
    const float l = sqrtf(mr_dot3(v, v)), inv = 1.0f / fmaxf(l, 1e-12f);
    if (l >= 1e-12f) {
        const float d = (v[0] * g[0] + v[1] * g[1] + v[2] * g[2]) * inv * inv;
        for (int k = 0; k < 3; ++k) out[k] = (g[k] - v[k] * d) * inv;
    } else {
        for (int k = 0; k < 3; ++k) out[k] = g[k] * inv;
    }
}

// face pass of the backward: g_fn = g_face_normals_ext[t] + sum over the face's vertices of g_sum (= d / d vn_sum, per vertex), then
// through normalize and the cross product to the three vertices (atomic adds into g_verts)
MVE_MR_FN void mr_normals_face_bwd(const float* verts, const int32_t* faces, int t, const float* g_fn_ext, const float* g_sum, float* g_verts) {
    const int32_t* f = faces + 3 * t;
    const float* a = verts + 3 * f[0];
    const float* b = verts + 3 * f[1];
    const float* c = verts + 3 * f[2];
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const float cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    float g[3], gc[3];
    for (int k = 0; k < 3; ++k) g[k] = (g_fn_ext ? g_fn_ext[3 * t + k] : 0.f) + g_sum[3 * f[0] + k] + g_sum[3 * f[1] + k] + g_sum[3 * f[2] + k];
    mr_normalize_bwd(cr, g, gc);
    // cr = e1 x e2:  g_e1 = e2 x gc,  g_e2 = gc x e1
    const float g1[3] = {e2[1] * gc[2] - e2[2] * gc[1], e2[2] * gc[0] - e2[0] * gc[2], e2[0] * gc[1] - e2[1] * gc[0]};
    const float g2[3] = {gc[1] * e1[2] - gc[2] * e1[1], gc[2] * e1[0] - gc[0] * e1[2], gc[0] * e1[1] - gc[1] * e1[0]};
    for (int k = 0; k < 3; ++k) {
        mr_add(g_verts + 3 * f[1] + k, g1[k]);
        mr_add(g_verts + 3 * f[2] + k, g2[k]);
        mr_add(g_verts + 3 * f[0] + k, -g1[k] - g2[k]);
    }
}
