// Occupancy-grid DDA shared by the ray-marching operators (raymarching.hip) and the fused NeRF renderer (nerf.hip).
// Arithmetic follows lib/ops/raymarching/src/raymarching.cu of the reference op for op (see raymarching.hip);
// every translation unit that includes this header must be compiled with -ffp-contract=off.
#pragma once
#include "common.h"
#include <float.h>

namespace {

constexpr float kSqrt3 = 1.7320508075688772f;

// ---------------------------------------------------------------------------
// integer helpers (bit exact)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread3(uint32_t v) {
    // 10 low bits of v -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
__device__ __forceinline__ uint32_t gather3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xC30C30C3u;
    x = (x | (x >> 4)) & 0x0F00F00Fu;
    x = (x | (x >> 8)) & 0xFF0000FFu;
    x = (x | (x >> 16)) & 0x0000FFFFu;
    return x;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

struct float3p { float x, y, z; };   // 12-byte packed load/store (one dwordx3 per lane)
struct float2p { float x, y; };

// ---------------------------------------------------------------------------
// The DDA shared by both marchers.
// ---------------------------------------------------------------------------
struct MarchParams {
    const uint8_t* grid;
    float bound;
    int contract;
    float dt_gamma;
    uint32_t max_steps;
    uint32_t C, H;
};

struct GridWalker {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float rH, H3, Hf, Cf, dt_min, dt_max, bound, dt_gamma;
    uint32_t H;
    int contract;
    const uint8_t* grid;

    __device__ __forceinline__ void init(const MarchParams& p, const float* o, const float* d) {
        const float3p O = *reinterpret_cast<const float3p*>(o);
        const float3p D = *reinterpret_cast<const float3p*>(d);
        ox = O.x; oy = O.y; oz = O.z;
        dx = D.x; dy = D.y; dz = D.z;
        rdx = 1 / dx; rdy = 1 / dy; rdz = 1 / dz;
        H = p.H;
        Hf = (float)p.H;
        Cf = (float)p.C;
        rH = 1 / (float)p.H;
        H3 = (float)(p.H * p.H * p.H);
        dt_min = 2 * kSqrt3 / p.max_steps;
        dt_max = 2 * kSqrt3 * p.bound / p.H;
        bound = p.bound;
        dt_gamma = p.dt_gamma;
        contract = p.contract;
        grid = p.grid;
    }

    __device__ __forceinline__ float step_len(float t) const { return clampf(t * dt_gamma, dt_min, dt_max); }

    __device__ __forceinline__ int cascade_of(float x, float y, float z, float dt) const {
        int e_pos, e_dt;
        const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        frexpf(mx, &e_pos);
        const int l_pos = (int)fminf(Cf - 1, fmaxf(0, e_pos));
        const float md = (float)(dt * Hf * 0.5);   // double multiply by the literal, as in the reference
        frexpf(md, &e_dt);
        const int l_dt = (int)fminf(Cf - 1, fmaxf(0, e_dt));
        return max(l_pos, l_dt);
    }

    // March from t; calls sink(cx,cy,cz,t_after,dt) -> bool (false = stop) for every occupied sample until
    // `budget` samples were produced or t >= far.  Returns #samples.
    template <class Sink>
    __device__ __forceinline__ uint32_t walk(float t, float far, uint32_t budget, Sink&& sink) const {
        uint32_t n = 0;
        while (t < far && n < budget) {
            const float x = clampf(ox + t * dx, -bound, bound);
            const float y = clampf(oy + t * dy, -bound, bound);
            const float z = clampf(oz + t * dz, -bound, bound);
            float dt = step_len(t);
            const int level = cascade_of(x, y, z, dt);
            const float mip_bound = fminf(scalbnf(1.0f, level), bound);
            const float mip_rbound = 1 / mip_bound;

            float cx = x, cy = y, cz = z;
            const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
            const bool warped = contract && mag > 1;
            if (warped) {
                const float s = (2 - 1 / mag) / mag;   // L-inf contraction
                cx *= s; cy *= s; cz *= s;
            }
            const float top = (float)(H - 1);
            const int nx = (int)clampf((float)(0.5 * (cx * mip_rbound + 1) * H), 0.0f, top);
            const int ny = (int)clampf((float)(0.5 * (cy * mip_rbound + 1) * H), 0.0f, top);
            const int nz = (int)clampf((float)(0.5 * (cz * mip_rbound + 1) * H), 0.0f, top);

            const uint32_t cell = (uint32_t)(level * H3 + morton_encode(nx, ny, nz));
            const bool occ = grid[cell >> 3] & (1u << (cell & 7u));

            if (occ) {
                t += dt;
                ++n;
                if (!sink(cx, cy, cz, t, dt)) break;    // the sink may end the march (fused renderer: ray saturated)
            } else if (warped) {
                t += dt;
            } else {
                // skip to the exit face of this voxel
                const float tx = (((nx + 0.5f + 0.5f * copysignf(1.0f, dx)) * rH * 2 - 1) * mip_bound - cx) * rdx;
                const float ty = (((ny + 0.5f + 0.5f * copysignf(1.0f, dy)) * rH * 2 - 1) * mip_bound - cy) * rdy;
                const float tz = (((nz + 0.5f + 0.5f * copysignf(1.0f, dz)) * rH * 2 - 1) * mip_bound - cz) * rdz;
                const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
                do {
                    dt = step_len(t);
                    t += dt;
                } while (t < tt);
            }
        }
        return n;
    }
};


// slab test of kernel_near_far_from_aabb (raymarching.cu:92-145), one ray
__device__ __forceinline__ void near_far_one(const float3p o, const float3p d, const float* __restrict__ aabb, float min_near,
                                             float& near_out, float& far_out) {
    const float lo[3] = {aabb[0], aabb[1], aabb[2]}, hi[3] = {aabb[3], aabb[4], aabb[5]};
    const float oo[3] = {o.x, o.y, o.z};
    const float rd[3] = {1 / d.x, 1 / d.y, 1 / d.z};
    float near = (lo[0] - oo[0]) * rd[0], far = (hi[0] - oo[0]) * rd[0];
    if (near > far) { float s = near; near = far; far = s; }
    bool miss = false;
#pragma unroll
    for (int a = 1; a < 3; ++a) {
        float na = (lo[a] - oo[a]) * rd[a], fa = (hi[a] - oo[a]) * rd[a];
        if (na > fa) { float s = na; na = fa; fa = s; }
        if (!miss) {
            if (near > fa || na > far) {
                miss = true;
            } else {
                if (na > near) near = na;
                if (fa < far) far = fa;
            }
        }
    }
    if (miss) {
        near = far = FLT_MAX;
    } else if (near < min_near) {
        near = min_near;
    }
    near_out = near;
    far_out = far;
}

}  // namespace
