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
