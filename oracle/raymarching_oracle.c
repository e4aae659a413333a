/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Sequential CPU restatement (plain C, one loop iteration per CUDA thread of
 * the reference) of the ray-marching kernels in
 *   lib/ops/raymarching/src/raymarching.cu            (Lakonik/MVEdit)
 * Each function cites the kernel it follows.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * Numerics: compiled with -O2 -ffp-contract=off -fno-fast-math so every
 * expression is evaluated exactly as the reference source spells it (float
 * ops in float, the `0.5 * ...` literals in double).  nvcc would additionally
 * contract some a*b+c into fma on the reference's own hardware; that is not
 * reproducible bit-for-bit on any other target and is not modelled.
 * `__expf` (fast exp) is restated with expf: compositing outputs are compared
 * with a relative tolerance of 1e-5, index buffers and marched samples bit-exact.
 *
 * Pinning: the reference ships no golden vectors for this path.
 * This restatement is pinned against the reference kernels themselves, built
 * for the GPU from the reference sources by oracle/build_ref.py into
 * oracle/_ref/ and compared in tests/test_raymarching_ref.py (gpu), and against
 * golden vectors generated from it under tests/golden/.
 *
 * Deterministic offsets: the reference obtains each ray's sample offset with
 * atomicAdd(counter, step) (raymarching.cu:471), i.e. in thread arrival order.
 * The oracle executes the threads in ray order, which is one legal
 * serialisation: offset[n] = sum_{m<n} count[m].
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define SQRT3F 1.7320508075688772f

static float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

/* raymarching.cu:56-63 */
static uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* raymarching.cu:65-71 */
static uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
/* raymarching.cu:73-81 */
static uint32_t morton3d_invert1(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* raymarching.cu:41-46 */
static int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, exponent));
}
/* raymarching.cu:48-53 */
static int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)(dt * H * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, exponent));
}

/* ---- kernel_near_far_from_aabb, raymarching.cu:92-145 ---- */
void orc_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                            float* nears, float* fars) {
    for (uint32_t n = 0; n < N; ++n) {
        const float* o = rays_o + (size_t)n * 3;
        const float* d = rays_d + (size_t)n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, tmp;
        if (near > far) { tmp = near; near = far; far = tmp; }
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { tmp = near_y; near_y = far_y; far_y = tmp; }
        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { tmp = near_z; near_z = far_z; far_z = tmp; }
        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* ---- kernel_morton3D / kernel_morton3D_invert, raymarching.cu:214-226, 237-254 ---- */
void orc_morton3d(const int32_t* coords, uint32_t N, int32_t* indices) {
    for (uint32_t n = 0; n < N; ++n)
        indices[n] = (int32_t)morton3d((uint32_t)coords[3 * (size_t)n], (uint32_t)coords[3 * (size_t)n + 1],
                                       (uint32_t)coords[3 * (size_t)n + 2]);
}
void orc_morton3d_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    for (uint32_t n = 0; n < N; ++n) {
        const int ind = indices[n];
        coords[3 * (size_t)n + 0] = (int32_t)morton3d_invert1((uint32_t)(ind >> 0));
        coords[3 * (size_t)n + 1] = (int32_t)morton3d_invert1((uint32_t)(ind >> 1));
        coords[3 * (size_t)n + 2] = (int32_t)morton3d_invert1((uint32_t)(ind >> 2));
    }
}

/* ---- kernel_packbits, raymarching.cu:268-289 ---- */
void orc_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
    for (uint32_t n = 0; n < N; ++n) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; ++i) bits |= (grid[(size_t)n * 8 + i] >= density_thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* ---- kernel_flatten_rays, raymarching.cu:303-319 ---- */
void orc_flatten_rays(const int32_t* rays, uint32_t N, uint32_t M, int32_t* res) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t offset = (uint32_t)rays[2 * (size_t)n], num_steps = (uint32_t)rays[2 * (size_t)n + 1];
        for (uint32_t i = 0; i < num_steps && offset + i < M; ++i) res[offset + i] = (int32_t)n;
    }
}

/*
 * The occupancy-grid DDA shared by kernel_march_rays_train (raymarching.cu:338-475)
 * and kernel_march_rays (:714-829): starting at t, produce at most `budget`
 * samples.  If xyzs is NULL only the count is returned (first pass).
 */
static uint32_t march_one(const float* o, const float* d, const uint8_t* grid, float bound, int contract, float dt_gamma,
                          uint32_t max_steps, uint32_t C, uint32_t H, float t, float far, uint32_t budget, float* xyzs,
                          float* dirs, float* ts) {
    const float ox = o[0], oy = o[1], oz = o[2];
    const float dx = d[0], dy = d[1], dz = d[2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    const float dt_min = 2 * SQRT3F / max_steps;
    const float dt_max = 2 * SQRT3F * bound / H;
    uint32_t step = 0;

    while (t < far && step < budget) {
        const float x = clampf(ox + t * dx, -bound, bound);
        const float y = clampf(oy + t * dy, -bound, bound);
        const float z = clampf(oz + t * dz, -bound, bound);
        float dt = clampf(t * dt_gamma, dt_min, dt_max);

        const int lp = mip_from_pos(x, y, z, (float)C), ld = mip_from_dt(dt, (float)H, (float)C);
        const int level = lp > ld ? lp : ld;
        const float mip_bound = fminf(scalbnf(1.0f, level), bound);
        const float mip_rbound = 1 / mip_bound;

        float cx = x, cy = y, cz = z;
        const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        if (contract && mag > 1) {
            const float Linf_scale = (2 - 1 / mag) / mag;
            cx *= Linf_scale; cy *= Linf_scale; cz *= Linf_scale;
        }
        const int nx = (int)clampf((float)(0.5 * (cx * mip_rbound + 1) * H), 0.0f, (float)(H - 1));
        const int ny = (int)clampf((float)(0.5 * (cy * mip_rbound + 1) * H), 0.0f, (float)(H - 1));
        const int nz = (int)clampf((float)(0.5 * (cz * mip_rbound + 1) * H), 0.0f, (float)(H - 1));

        const uint32_t index = (uint32_t)(level * H3 + morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        const int occ = grid[index / 8] & (1 << (index % 8));

        if (occ) {
            t += dt;
            if (xyzs) {
                xyzs[0] = cx; xyzs[1] = cy; xyzs[2] = cz;
                dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
                ts[0] = t; ts[1] = dt;
                xyzs += 3; dirs += 3; ts += 2;
            }
            step++;
        } else if (contract && mag > 1) {
            t += dt;
        } else {
            const float tx = (((nx + 0.5f + 0.5f * copysignf(1.0f, dx)) * rH * 2 - 1) * mip_bound - cx) * rdx;
            const float ty = (((ny + 0.5f + 0.5f * copysignf(1.0f, dy)) * rH * 2 - 1) * mip_bound - cy) * rdy;
            const float tz = (((nz + 0.5f + 0.5f * copysignf(1.0f, dz)) * rH * 2 - 1) * mip_bound - cz) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do {
                dt = clampf(t * dt_gamma, dt_min, dt_max);
                t += dt;
            } while (t < tt);
        }
    }
    return step;
}

static float start_t(float t, float dt_gamma, uint32_t max_steps, float bound, uint32_t H, float noise) {
    const float dt_min = 2 * SQRT3F / max_steps;
    const float dt_max = 2 * SQRT3F * bound / H;
    t += clampf(t * dt_gamma, dt_min, dt_max) * noise;
    return t;
}

/* ---- kernel_march_rays_train, first pass (xyzs == nullptr), raymarching.cu:338-475.
 * Returns M.  rays[n] = (offset, count) with offsets in ray order. ---- */
uint32_t orc_march_rays_train_count(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                                    float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                    const float* nears, const float* fars, const float* noises, int32_t* rays) {
    uint32_t counter = 0;
    for (uint32_t n = 0; n < N; ++n) {
        const float t0 = start_t(nears[n], dt_gamma, max_steps, bound, H, noises[n]);
        const uint32_t step = march_one(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, contract, dt_gamma,
                                        max_steps, C, H, t0, fars[n], max_steps, NULL, NULL, NULL);
        rays[2 * (size_t)n] = (int32_t)counter;   /* atomicAdd(counter, step) executed in ray order */
        rays[2 * (size_t)n + 1] = (int32_t)step;
        counter += step;
    }
    return counter;
}

/* ---- second pass, raymarching.cu:361-367 + loop ---- */
void orc_march_rays_train_write(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                                float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float* nears,
                                const float* fars, const float* noises, const int32_t* rays, float* xyzs, float* dirs,
                                float* ts) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t point_index = (uint32_t)rays[2 * (size_t)n], num_steps = (uint32_t)rays[2 * (size_t)n + 1];
        const float t0 = start_t(nears[n], dt_gamma, max_steps, bound, H, noises[n]);
        march_one(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid, bound, contract, dt_gamma, max_steps, C, H, t0,
                  fars[n], num_steps, xyzs + 3 * (size_t)point_index, dirs + 3 * (size_t)point_index,
                  ts + 2 * (size_t)point_index);
    }
}

/* ---- kernel_composite_rays_train_forward, raymarching.cu:501-579 ---- */
void orc_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays, uint32_t M,
                                      uint32_t N, float T_thresh, int binarize, float* weights, float* weights_sum,
                                      float* depth, float* image) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t offset = (uint32_t)rays[2 * (size_t)n], num_steps = (uint32_t)rays[2 * (size_t)n + 1];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[n] = 0; depth[n] = 0;
            image[3 * (size_t)n] = image[3 * (size_t)n + 1] = image[3 * (size_t)n + 2] = 0;
            continue;
        }
        const float* t = ts + 2 * (size_t)offset;
        const float* s = sigmas + offset;
        const float* c = rgbs + 3 * (size_t)offset;
        float* w = weights + offset;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
        for (uint32_t step = 0; step < num_steps; ++step) {
            const float real_alpha = 1.0f - expf(-s[0] * t[1]);
            const float alpha = binarize ? (real_alpha > 0.5 ? 1.0 : 0.0) : real_alpha;
            const float weight = alpha * T;
            w[0] = weight;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            ws += weight;
            d += weight / t[0];
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            w++; s++; c += 3; t += 2;
        }
        weights_sum[n] = ws; depth[n] = d;
        image[3 * (size_t)n] = r; image[3 * (size_t)n + 1] = g; image[3 * (size_t)n + 2] = b;
    }
}

/* ---- kernel_composite_rays_train_backward, raymarching.cu:606-695 ---- */
void orc_composite_rays_train_backward(const float* grad_weights, const float* grad_weights_sum, const float* grad_depth,
                                       const float* grad_image, const float* sigmas, const float* rgbs, const float* ts,
                                       const int32_t* rays, const float* weights_sum, const float* depth, const float* image,
                                       uint32_t M, uint32_t N, float T_thresh, int binarize, float* grad_sigmas,
                                       float* grad_rgbs) {
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t offset = (uint32_t)rays[2 * (size_t)n], num_steps = (uint32_t)rays[2 * (size_t)n + 1];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float* gw = grad_weights + offset;
        const float gws = grad_weights_sum[n], gd = grad_depth[n];
        const float* gi = grad_image + 3 * (size_t)n;
        const float* s = sigmas + offset;
        const float* c = rgbs + 3 * (size_t)offset;
        const float* t = ts + 2 * (size_t)offset;
        float* gs = grad_sigmas + offset;
        float* gc = grad_rgbs + 3 * (size_t)offset;
        const float r_final = image[3 * (size_t)n], g_final = image[3 * (size_t)n + 1], b_final = image[3 * (size_t)n + 2];
        const float ws_final = weights_sum[n], d_final = depth[n];
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
        for (uint32_t step = 0; step < num_steps; ++step) {
            const float real_alpha = 1.0f - expf(-s[0] * t[1]);
            const float alpha = binarize ? (real_alpha > 0.5 ? 1.0 : 0.0) : real_alpha;
            const float weight = alpha * T;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            ws += weight;
            d += weight / t[0];
            T *= 1.0f - alpha;
            gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
            gs[0] = t[1] * (gi[0] * (T * c[0] - (r_final - r)) + gi[1] * (T * c[1] - (g_final - g)) +
                            gi[2] * (T * c[2] - (b_final - b)) + (gws + gw[0]) * (T - (ws_final - ws)) +
                            gd * (T / t[0] - (d_final - d)));
            if (T < T_thresh) break;
            s++; c += 3; t += 2; gw++; gs++; gc += 3;
        }
    }
}

/* ---- kernel_march_rays (inference), raymarching.cu:714-829 ---- */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                    const float* rays_d, float bound, int contract, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                    const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* ts,
                    const float* noises) {
    (void)nears;
    for (uint32_t n = 0; n < n_alive; ++n) {
        const int index = rays_alive[n];
        const float t0 = start_t(rays_t[index], dt_gamma, max_steps, bound, H, noises[n]);
        march_one(rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, grid, bound, contract, dt_gamma, max_steps, C, H, t0,
                  fars[index], n_step, xyzs + 3 * (size_t)n * n_step, dirs + 3 * (size_t)n * n_step,
                  ts + 2 * (size_t)n * n_step);
    }
}

/* ---- kernel_composite_rays (inference), raymarching.cu:843-925 ---- */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int binarize, int32_t* rays_alive, float* rays_t,
                        const float* sigmas, const float* rgbs, const float* ts, float* weights_sum, float* depth,
                        float* image) {
    for (uint32_t n = 0; n < n_alive; ++n) {
        const int index = rays_alive[n];
        const float* s = sigmas + (size_t)n * n_step;
        const float* c = rgbs + 3 * (size_t)n * n_step;
        const float* tp = ts + 2 * (size_t)n * n_step;
        float t = 0;
        float d = depth[index], r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1],
              b = image[3 * (size_t)index + 2], weight_sum = weights_sum[index];
        uint32_t step = 0;
        while (step < n_step) {
            if (tp[0] == 0) break;
            const float real_alpha = 1.0f - expf(-s[0] * tp[1]);
            const float alpha = binarize ? (real_alpha > 0.5 ? 1.0 : 0.0) : real_alpha;
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t = tp[0];
            d += weight / t;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            if (T < T_thresh) break;
            s++; c += 3; tp += 2; step++;
        }
        if (step < n_step) rays_alive[n] = -1;
        else rays_t[index] = t;
        weights_sum[index] = weight_sum;
        depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
    }
}
