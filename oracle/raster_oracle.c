/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Reference software rasteriser + attribute interpolation with the OUTPUT CONVENTION of nvdiffrast's
 * dr.rasterize / dr.interpolate, as consumed by the reference at
 *   lib/models/decoders/mesh_renderer/base_mesh_renderer.py:240-252 ("[u, v, z/w, triangle_id]", fg <=> rast[...,3] > 0).
 * nvdiffrast itself (requirements.txt:3, commit c5caf7b) is third-party and absent from /root/reference and from this
 * image, so its exact coverage/tie rules cannot be consulted: PARITY UNPINNED.  The rules below are this repo's
 * specification; the HIP rasteriser (mvedit_amd/csrc/raster.hip) must reproduce them bit-for-bit:
 *   - clip-space input pos[B][V][4], triangles tri[F][3]; a triangle with any w <= 1e-6 is skipped (no near-plane
 *     clipping; the MVEdit rigs keep the object 3.7 units in front of a near plane at 0.01);
 *   - screen position sx = (x/w * 0.5 + 0.5) * W, sy likewise with H, in float32, snapped to 1/256 pixel:
 *     X = (int)floorf(sx * 256 + 0.5f); pixel (px,py) is sampled at (256 px + 128, 256 py + 128); row 0 is y_ndc = -1
 *     (OpenGL orientation, as nvdiffrast; the reference flips y in its projection matrix, base_mesh_renderer.py:229);
 *   - coverage by exact int64 edge functions, both windings, top-left fill rule on ties, zero-area triangles skipped;
 *   - screen-space barycentrics b_i = E_i / (E_0+E_1+E_2) in float32; depth z/w = sum b_i z_i/w_i must lie in [-1, 1];
 *     nearest depth wins, equal depth -> lower triangle index;
 *   - rast = (u, v, z/w, id+1) with perspective-correct u = (b0/w0)/S, v = (b1/w1)/S, S = sum b_i/w_i; empty pixel = 0.
 * Built with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int64_t edge_fn(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py) {
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
/* top-left rule for an edge a->b of a positively oriented triangle: ties belong to the triangle on "left" and "top" edges */
static inline int edge_owns_tie(int64_t ax, int64_t ay, int64_t bx, int64_t by) {
    const int64_t dx = bx - ax, dy = by - ay;
    return (dy > 0) || (dy == 0 && dx < 0);
}

void orc_rasterize(const float* pos, int B, int V, const int32_t* tri, int F, int H, int W, float* rast) {
    const size_t npix = (size_t)H * W;
    float* zbuf = (float*)malloc(npix * sizeof(float));
    for (int b = 0; b < B; ++b) {
        float* out = rast + (size_t)b * npix * 4;
        memset(out, 0, npix * 4 * sizeof(float));
        for (size_t i = 0; i < npix; ++i) zbuf[i] = 2.0f;
        const float* P = pos + (size_t)b * V * 4;
        for (int f = 0; f < F; ++f) {
            const int32_t i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
            if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= V || i1 >= V || i2 >= V) continue;
            const float* v[3] = {P + 4 * i0, P + 4 * i1, P + 4 * i2};
            if (v[0][3] <= 1e-6f || v[1][3] <= 1e-6f || v[2][3] <= 1e-6f) continue;
            int64_t X[3], Y[3];
            float zw[3], iw[3];
            for (int k = 0; k < 3; ++k) {
                const float sx = (v[k][0] / v[k][3] * 0.5f + 0.5f) * (float)W;
                const float sy = (v[k][1] / v[k][3] * 0.5f + 0.5f) * (float)H;
                X[k] = (int64_t)floorf(sx * 256.0f + 0.5f);
                Y[k] = (int64_t)floorf(sy * 256.0f + 0.5f);
                zw[k] = v[k][2] / v[k][3];
                iw[k] = 1.0f / v[k][3];
            }
            int64_t area = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
            if (area == 0) continue;
            const int64_t sgn = area > 0 ? 1 : -1;
            int64_t xmin = X[0], xmax = X[0], ymin = Y[0], ymax = Y[0];
            for (int k = 1; k < 3; ++k) {
                if (X[k] < xmin) xmin = X[k]; if (X[k] > xmax) xmax = X[k];
                if (Y[k] < ymin) ymin = Y[k]; if (Y[k] > ymax) ymax = Y[k];
            }
            /* pixel centres 256 p + 128 inside [min, max] */
            int64_t px0 = (xmin - 128 + 255) >> 8, px1 = (xmax - 128) >> 8, py0 = (ymin - 128 + 255) >> 8, py1 = (ymax - 128) >> 8;
            if (px0 < 0) px0 = 0; if (py0 < 0) py0 = 0;
            if (px1 > W - 1) px1 = W - 1; if (py1 > H - 1) py1 = H - 1;
            /* tie ownership per edge, for the positively oriented version of the triangle */
            int own[3];
            for (int k = 0; k < 3; ++k) {
                const int a = (k + 1) % 3, c = (k + 2) % 3;      /* edge opposite vertex k: a -> c */
                own[k] = sgn > 0 ? edge_owns_tie(X[a], Y[a], X[c], Y[c]) : edge_owns_tie(X[c], Y[c], X[a], Y[a]);
            }
            for (int64_t py = py0; py <= py1; ++py)
                for (int64_t px = px0; px <= px1; ++px) {
                    const int64_t cx = px * 256 + 128, cy = py * 256 + 128;
                    int64_t E[3];
                    int inside = 1;
                    for (int k = 0; k < 3; ++k) {
                        const int a = (k + 1) % 3, c = (k + 2) % 3;
                        E[k] = sgn * edge_fn(X[a], Y[a], X[c], Y[c], cx, cy);
                        if (E[k] < 0 || (E[k] == 0 && !own[k])) { inside = 0; break; }
                    }
                    if (!inside) continue;
                    const float tot = (float)(E[0] + E[1] + E[2]);
                    const float b0 = (float)E[0] / tot, b1 = (float)E[1] / tot, b2 = (float)E[2] / tot;
                    const float z = b0 * zw[0] + b1 * zw[1] + b2 * zw[2];
                    if (!(z >= -1.0f && z <= 1.0f)) continue;
                    const size_t pi = (size_t)py * W + px;
                    if (z < zbuf[pi]) {                 /* triangles are visited in index order: equal depth keeps the lower id */
                        zbuf[pi] = z;
                        const float q0 = b0 * iw[0], q1 = b1 * iw[1], q2 = b2 * iw[2];
                        const float S = q0 + q1 + q2;
                        out[4 * pi + 0] = q0 / S;
                        out[4 * pi + 1] = q1 / S;
                        out[4 * pi + 2] = z;
                        out[4 * pi + 3] = (float)(f + 1);
                    }
                }
        }
    }
    free(zbuf);
}

/* dr.interpolate: attr[Battr][Vattr][A] (Battr = 1 broadcasts), rast[B][H*W][4], tri[F][3] -> out[B][H*W][A]; empty pixels 0 */
void orc_interpolate(const float* attr, int Battr, int Vattr, int A, const float* rast, int B, int npix, const int32_t* tri, int F,
                     float* out) {
    for (int b = 0; b < B; ++b) {
        const float* at = attr + (Battr > 1 ? (size_t)b * Vattr * A : 0);
        for (int i = 0; i < npix; ++i) {
            const float* r = rast + ((size_t)b * npix + i) * 4;
            float* o = out + ((size_t)b * npix + i) * A;
            const int id = (int)r[3] - 1;
            if (id < 0 || id >= F) { for (int a = 0; a < A; ++a) o[a] = 0.0f; continue; }
            const float u = r[0], v = r[1], w = 1.0f - u - v;
            const float* a0 = at + (size_t)tri[3 * id] * A;
            const float* a1 = at + (size_t)tri[3 * id + 1] * A;
            const float* a2 = at + (size_t)tri[3 * id + 2] * A;
            for (int a = 0; a < A; ++a) o[a] = u * a0[a] + v * a1[a] + w * a2[a];
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * dr.antialias as consumed at base_mesh_renderer.py:289-293 (rgba | depth | normal, 8 channels, one mesh).
 * nvdiffrast's implementation is not available (see the header of this file): PARITY UNPINNED.  The rules below are this
 * repo's specification of the published idea (blend across silhouette edges by the sub-pixel position of the edge):
 *   - opp[F][3]: for edge e = (tri[e], tri[(e+1)%3]) of a triangle, the vertex opposite to it in the ONE other triangle
 *     that shares the same undirected edge; -1 when the edge is used by 1 or by more than 2 triangles;
 *   - for every pair of 4-adjacent pixels (P, Q) whose triangle ids differ: take the triangle that is in front (background
 *     loses, then smaller z/w, ties to the pixel with the smaller linear index); P := the pixel that shows it, Q := the other;
 *   - an edge (a, b) of that triangle with third vertex c is a silhouette iff opp < 0 or the opposite vertex lies on the same
 *     side of line ab as c in screen space (cross products with >= 0);
 *   - screen positions are the float32 (x/w*0.5+0.5)*W of the rasteriser, pixel centres at +0.5; the crossing of the line ab
 *     with the axis-aligned segment P->Q is s in [0,1] along ab and t in [0,1] from P to Q; the smallest t over the
 *     silhouette edges wins; no crossing -> no blend;
 *   - t > 0.5: Q += (t - 0.5) (in[P] - in[Q]);   t < 0.5: P += (0.5 - t) (in[Q] - in[P]);
 *   - each output pixel gathers its (up to four) contributions in the fixed order left, right, up, down from the INPUT image.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { uint64_t key; int ent; } edge_rec;
static int edge_cmp(const void* a, const void* b) {
    const edge_rec *x = (const edge_rec*)a, *y = (const edge_rec*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->ent - y->ent;
}
void orc_edge_opposites(const int32_t* tri, int F, int32_t* opp) {
    edge_rec* r = (edge_rec*)malloc(sizeof(edge_rec) * 3 * (size_t)F);
    for (int f = 0; f < F; ++f)
        for (int e = 0; e < 3; ++e) {
            const uint32_t a = (uint32_t)tri[3 * f + e], b = (uint32_t)tri[3 * f + (e + 1) % 3];
            r[3 * f + e].key = a < b ? ((uint64_t)a << 32 | b) : ((uint64_t)b << 32 | a);
            r[3 * f + e].ent = 3 * f + e;
        }
    qsort(r, 3 * (size_t)F, sizeof(edge_rec), edge_cmp);
    for (size_t i = 0; i < 3 * (size_t)F;) {
        size_t j = i;
        while (j < 3 * (size_t)F && r[j].key == r[i].key) ++j;
        for (size_t k = i; k < j; ++k) {
            int o = -1;
            if (j - i == 2) {
                const int other = r[k == i ? i + 1 : i].ent;
                o = tri[3 * (other / 3) + (other % 3 + 2) % 3];
            }
            opp[r[k].ent] = o;
        }
        i = j;
    }
    free(r);
}

/* blend weight and direction of the pair (pixel p, neighbour q = p + (dx,dy)); returns 1 and (*dst_is_p, *wgt) when a blend happens */
static int aa_pair(const float* rast, const float* pos, int V, const int32_t* tri, int F, const int32_t* opp, int H, int W, int px,
                   int py, int qx, int qy, int* dst_is_p, float* wgt) {
    const float* rp = rast + ((size_t)py * W + px) * 4;
    const float* rq = rast + ((size_t)qy * W + qx) * 4;
    const int ip = (int)rp[3] - 1, iq = (int)rq[3] - 1;
    if (ip == iq) return 0;
    int use_p;
    if (ip < 0) use_p = 0;
    else if (iq < 0) use_p = 1;
    else if (rp[2] != rq[2]) use_p = rp[2] < rq[2];
    else use_p = (py * W + px) < (qy * W + qx);
    const int t = use_p ? ip : iq;
    if (t < 0 || t >= F) return 0;
    const int ox = use_p ? px : qx, oy = use_p ? py : qy;          /* owner pixel P of the front triangle */
    const int nx = use_p ? qx : px, ny = use_p ? qy : py;          /* the other pixel Q */
    float sx[3], sy[3];
    int vi[3];
    for (int k = 0; k < 3; ++k) {
        vi[k] = tri[3 * t + k];
        if (vi[k] < 0 || vi[k] >= V) return 0;
        const float* v = pos + 4 * (size_t)vi[k];
        if (v[3] <= 1e-6f) return 0;
        sx[k] = (v[0] / v[3] * 0.5f + 0.5f) * (float)W;
        sy[k] = (v[1] / v[3] * 0.5f + 0.5f) * (float)H;
    }
    const float cx = (float)ox + 0.5f, cy = (float)oy + 0.5f;
    const float dx = (float)(nx - ox), dy = (float)(ny - oy);
    float best = 2.0f;
    for (int e = 0; e < 3; ++e) {
        const int a = e, b = (e + 1) % 3, c = (e + 2) % 3;
        const float ex = sx[b] - sx[a], ey = sy[b] - sy[a];
        const int o = opp[3 * t + e];
        if (o >= 0) {
            if (o >= V) continue;
            const float* v = pos + 4 * (size_t)o;
            if (v[3] <= 1e-6f) continue;
            const float oxs = (v[0] / v[3] * 0.5f + 0.5f) * (float)W, oys = (v[1] / v[3] * 0.5f + 0.5f) * (float)H;
            const float sc = ex * (sy[c] - sy[a]) - ey * (sx[c] - sx[a]);
            const float so = ex * (oys - sy[a]) - ey * (oxs - sx[a]);
            if (!(sc * so >= 0.0f)) continue;                       /* neighbour continues the surface: not a silhouette */
        }
        float s, tt;
        if (dy == 0.0f) {                                           /* horizontal pair: cross the line y = cy */
            if (ey == 0.0f) continue;
            s = (cy - sy[a]) / ey;
            tt = ((sx[a] + s * ex) - cx) * dx;
        } else {
            if (ex == 0.0f) continue;
            s = (cx - sx[a]) / ex;
            tt = ((sy[a] + s * ey) - cy) * dy;
        }
        if (s >= 0.0f && s <= 1.0f && tt >= 0.0f && tt <= 1.0f && tt < best) best = tt;
    }
    if (best > 1.0f) return 0;
    if (best > 0.5f) { *dst_is_p = !use_p; *wgt = best - 0.5f; }    /* the far pixel Q receives */
    else if (best < 0.5f) { *dst_is_p = use_p; *wgt = 0.5f - best; }
    else return 0;
    return 1;
}

void orc_antialias(const float* color, int B, int H, int W, int C, const float* rast, const float* pos, int V, const int32_t* tri,
                   int F, const int32_t* opp, float* out) {
    static const int ddx[4] = {-1, 1, 0, 0}, ddy[4] = {0, 0, -1, 1};
    for (int b = 0; b < B; ++b) {
        const float* col = color + (size_t)b * H * W * C;
        const float* ra = rast + (size_t)b * H * W * 4;
        const float* po = pos + (size_t)b * V * 4;
        float* o = out + (size_t)b * H * W * C;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float* self = col + ((size_t)y * W + x) * C;
                float* dst = o + ((size_t)y * W + x) * C;
                for (int c = 0; c < C; ++c) dst[c] = self[c];
                for (int k = 0; k < 4; ++k) {
                    const int qx = x + ddx[k], qy = y + ddy[k];
                    if (qx < 0 || qx >= W || qy < 0 || qy >= H) continue;
                    int dst_is_p;
                    float w;
                    if (!aa_pair(ra, po, V, tri, F, opp, H, W, x, y, qx, qy, &dst_is_p, &w) || !dst_is_p) continue;
                    const float* nb = col + ((size_t)qy * W + qx) * C;
                    for (int c = 0; c < C; ++c) dst[c] = dst[c] + w * (nb[c] - self[c]);
                }
            }
    }
}
