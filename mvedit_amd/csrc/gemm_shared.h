// Definitions shared by the GEMM / implicit-GEMM conv kernels (gemm.hip: 128 x 160 tiles; gemm_big.hip: 256 x 320 tiles).  Everything is in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"

namespace {

constexpr int BK = 64;           // elements
constexpr int NT = 256;
constexpr int ROW_BYTES = BK * 2;  // 128 B per tile row

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

struct ConvGeom {
    // virtual input (after optional upsample) Hv x Wv, source tensors Hs x Ws
    int Hs, Ws, Hv, Wv, Ho, Wo;
    int C1, C2;        // channels of source 1 / source 2 (C2 = 0: no concat)
    int stride;        // 1 or 2
    int pad;           // rows / columns of zero padding before the first pixel: 1 (symmetric), 0 (MVE_CONV_PAD_BR: bottom/right only)
    int ups;           // 0 or 1 (nearest 2x)
    int chunk64;       // 1: K order is (channel slab of 64, tap, channel in slab)
    // fused 1x1 shortcut (ResnetBlock2D conv2 + conv_shortcut in ONE K loop): after the nk_main tiles of the 3x3 part, K continues
    // over the channels of up to two more NHWC tensors sampled at the output pixel (centre tap); 0 = no shortcut part
    int nk_main, C3, C4;
    // window: 0 / 3 = the 3 x 3 taps; 2 = a 2 x 2 window (K = 4 C, taps in row-major order) whose origin is (y - pad, x - pad_x): one output-parity
    // phase of nearest-2x upsample + 3 x 3 conv, see mve_upsample_conv_phases.  2 x 2 windows: ping-pong kernel (gemm_pp.hip) only.
    int kw;
    int pad_x;         // columns of zero padding before the first pixel (= pad for every 3 x 3 conv)
    // phase_rows = R > 0 (2 x 2 windows only): ONE launch computes all four phases -- its M = 4 R rows are [phase][image][y][x], rows
    // [ph R, (ph + 1) R) take pad = 1 - (ph >> 1), pad_x = 1 - (ph & 1) and the weights W + ph N ldw, and write through the grouped output rows
    // shifted by ((ph >> 1) * orow_extra + (ph & 1) * ldc / 2).  R is a multiple of 256: a tile never straddles two phases.
    int phase_rows;
};

// phase of row m in a phase_rows launch (0 otherwise) -- three compares, no division
__device__ __forceinline__ int conv_phase_of(int m, int R) { return R > 0 ? (int)(m >= R) + (int)(m >= 2 * R) + (int)(m >= 3 * R) : 0; }

struct GemmParams {
    const void* A;     // dense A [M][lda]  or conv source 1 (NHWC)
    const void* A2;    // conv source 2 (concat) or null
    const void* A3;    // shortcut source 1 (see ConvGeom::nk_main) or null
    const void* A4;    // shortcut source 2 or null
    const void* W;     // [N][K]
    void* out;         // [M][ldc] (or [M][ldc] with N/2 valid columns for GEGLU)
    const float* bias;       // [N] or null
    const float* rowvec;     // [M/rows_per_vec][ldrv] f32 (time embedding), or null
    const void* residual;    // [M][ldr] 16-bit or null
    // Residual stream as an unrounded pair (round 4, the executor's `residual_pair` mode; the default since round 5): a stream tensor x is stored as
    // hi = round16(x) -- what every MFMA operand read sees -- and an 8-bit low half lo8 = E5M2(2^8 (x - hi)) (common.h: mve_lo8_*; 16 bits in round 4).
    // Both null: the single 16-bit tensors of rounds 1-3.
    const void* residual_lo; // [M][ldr] BYTES: the low half of the residual, or null (residual is then the whole value)
    void* out_lo;            // [M][ldc] BYTES: receives lo8(v - round16(v)) next to out = round16(v), or null
    int M, N, K;
    int lda, ldw, ldc, ldr, ldrv;   // row strides (elements) of A, W, out, residual, rowvec
    int rows_per_vec;
    int geglu;         // 1: out[m][i] = v[2i] * gelu(v[2i+1])
    int out_f32;       // 1: out is float
    float out_scale;   // multiplies the final value (1/output_scale_factor)
    int res_after_scale;   // 1: out = (acc + bias) * out_scale + residual  (ControlNet: running sum over nets of scale * zero_conv)
    int splitk;        // > 1: K is cut into `splitk` slices, each block writes an fp32 partial tile to `partial`
    float* partial;    // [splitk][M][N] fp32 workspace (split-K only)
    int splitk_seq;    // > 1 (big-tile kernel only): ONE block walks all of K but rounds like `splitk_seq` concurrent slices --
                       // at every slice boundary the accumulators are folded into a block-private fp32 running total kept in
                       // `partial`, so the result is bit-identical to split-K + reducer without the reducer's traffic / launch
    int tile_n;        // 256-row ping-pong tile only: 0 = widest width that divides N (320 / 256 / 128); 160 = the 160-wide tile (N % 160 == 0),
                       // which the dispatcher picks when the 320-wide tiling would leave CUs without a block (small batches)
    int old_swizzle;   // ping-pong tile only, A/B aid: 1 = round 2's ring swizzle (2-way bank conflicts on every fragment read)
    int dbg;           // development switch (MVE_PP_DBG, 0 in every shipped path): bit 1 = pair launches of the 256-row tile take the generic epilogue path (A/B aid)
    // Output rows in groups: row m of out (and out_lo) starts at element m * ldc + (m >> orow_shift) * orow_extra.  orow_extra = 0: plain rows.
    // The phase convs of mve_upsample_conv_phases write pixel (b, i, j) of a [B][H][W] launch to pixel (b, 2 i + py, 2 j + px) of the [B][2H][2W]
    // NHWC output: ldc = 2 C, orow_shift = log2 W, orow_extra = 2 W C, base shifted by (py 2 W + px) C.
    int orow_shift, orow_extra;
    // In-kernel slice reduction (round 6, ping-pong tile, splitk > 1): one zero-initialised counter per output tile.  Non-null: the S blocks of a tile
    // write their fp32 partials, count themselves in, wait until all S have arrived, and each folds ITS share of the tile's rows over the slices
    // in slice order 0 .. S - 1 and runs the fused epilogue on it -- the arithmetic of k_splitk_reduce, bit for bit, without the second launch.
    // The last block to finish puts the counter back to zero.
    int* sk_sync;
    // Block -> tile order (round 6).  0: row panel major -- consecutive logical blocks (one XCD, one L2) share an activation row panel and walk the
    // weight strips: every XCD reads 1/8 of the activations and ALL of W.  1: weight strip major -- consecutive blocks share a (column tile, K slice)
    // strip of W and walk the row panels: every XCD reads 1/8 of W and all of the activations.  The dispatcher sets it where the activations are
    // the smaller operand (few rows: the deep UNet levels, small batches), where re-fetching all of W into eight L2s over the fabric is what
    // bounds the launch (profiles/r06_ab_deep_ring_null.log: 118-236 MB of weight re-reads per 8-image conv).  Pure index remap: identical bits.
    int w_major;
    // LayerNorm of the produced rows inside the launch (round 6; mve_gemm_pair_ln): ln_out [M][ld_ln] receives LayerNorm(out row) * gamma + beta in the
    // storage type, computed from the value the stand-alone kernel would read back (hi + lo8 of the row just written) with the shared row arithmetic
    // of ln_core.h -- bit-identical to mve_layernorm_pair on the output.  Only the 320-wide pair fast path of the ping-pong tile does it (N = 320:
    // a tile holds whole rows); every other launch leaves ln_out untouched and the caller runs the kernel (gemm.hip: mve_gemm_pair_ln).
    void* ln_out;
    const float* ln_gamma;
    const float* ln_beta;
    float ln_eps;
    int ld_ln;
    ConvGeom g;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

// erf-GELU x * Phi(x) with Phi(x) = 0.5 + t P(t^2) / Q(t^2), t = clamp(x, -5, 5): a (3, 3) rational minimax fit (tools/fit_gelu.py) whose
// absolute error is <= 1.1e-5 over all x when evaluated in fp32 -- 1/45 of the fp16 rounding step of an output of magnitude 1 -- for 11 plain
// VALU operations and one reciprocal.  The GEGLU epilogue evaluates 3.4e8 of these per level-0 launch and is VALU-bound: the previous form
// (Abramowitz & Stegun 7.1.26 through rcp + exp, 16 plain operations + 2 transcendentals, |error| <= 1.5e-7) cost 0.19 ms of a 0.78 ms launch
// (profiles/r03_ab_gelu.log).  Q >= 1 everywhere: the reciprocal is safe.
__device__ __forceinline__ float gelu_erf(float x) {
    const float t = __builtin_amdgcn_fmed3f(x, -5.0f, 5.0f);
    const float u = t * t;
    const float P = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.99665181e-05f, u, 0.00375327937f), u, 0.0294303672f), u, 0.398879537f);
    const float Q = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.00109351574f, u, 0.0246359949f), u, 0.240117083f), u, 1.0f);
    return x * __builtin_fmaf(t * P, __builtin_amdgcn_rcpf(Q), 0.5f);
}

// address of the 16-byte source chunk of im2col element (row = output pixel (cb,cy,cx), tap, channel `cin`)
template <class T>
__device__ __forceinline__ const T* conv_src(const GemmParams& p, int cb, int cy, int cx, int tap, int cin, bool kin) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const int yi = cy + dy, xi = cx + dx;
    const bool ok = kin && yi >= 0 && yi < p.g.Hv && xi >= 0 && xi < p.g.Wv;
    if (!ok) return nullptr;
    const bool second = cin >= p.g.C1;
    const T* src = reinterpret_cast<const T*>(second ? p.A2 : p.A);
    const int cs = second ? p.g.C2 : p.g.C1;
    const int ch = second ? cin - p.g.C1 : cin;
    const int ys = yi >> p.g.ups, xs = xi >> p.g.ups;
    return src + (((size_t)cb * p.g.Hs + ys) * p.g.Ws + xs) * cs + ch;
}


// Last part of the fused epilogue with the residual chunk already loaded (the 256-row tiles request it one pass ahead, see
// gemm_big_epilogue.h): residual, output scale, storage-dtype (or fp32) store.  One definition, contraction off, so that every kernel
// and the split-K reducer round identically.
// PAIR = false compiles the residual-pair handling out (the 256-row kernels that never see a pair keep the register allocation of rounds 1-3).
// RES_DONE: the caller's accumulators started from the residual (the PAIR instantiation of k_gemm_pp): nothing to add here.
template <class Tag, bool PAIR = true, bool RES_DONE = false>
__device__ __forceinline__ void gemm_epilogue_tail(const GemmParams& p, int m, int n, float (&v)[8], const typename Tag::V8& rr, bool ok = true) {
#pragma clang fp contract(off)
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    if (!RES_DONE && p.residual && !p.res_after_scale) {
        if (PAIR && p.residual_lo) {     // the pair is added as ONE value: hi + lo is exact in fp32 (|lo| <= ulp(hi) / 2)
            const u32x2 rl = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(p.residual_lo) + (size_t)m * p.ldr + n);
            float r8[8];
            mve_pair_load8<Tag>(rr, rl, r8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r8[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += Tag::to_f32(rr[e]);
        }
    }
    if (p.out_scale != 1.0f) {           // (x * 1 == x bit for bit: the test only saves the multiplies)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
    }
    if (!RES_DONE && p.residual && p.res_after_scale) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += Tag::to_f32(rr[e]);
    }
    // (`ok` guards only the stores: callers that keep loads in flight across chunks must not wrap the arithmetic in divergent control flow)
    size_t orow;
    if (p.g.phase_rows > 0) {            // (uniform branch: the four phases of mve_upsample_conv_phases in one launch)
        const int ph = conv_phase_of(m, p.g.phase_rows), mm = m - ph * p.g.phase_rows;
        orow = (size_t)mm * p.ldc + (size_t)((mm >> p.orow_shift) + (ph >> 1)) * p.orow_extra + (size_t)(ph & 1) * (p.ldc >> 1);
    } else orow = (size_t)m * p.ldc + (size_t)(m >> p.orow_shift) * p.orow_extra;
    if (p.out_f32) {
        float* op = reinterpret_cast<float*>(p.out) + orow + n;
        if (ok) {
            *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    } else {
        V8 pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = Tag::from_f32(v[e]);
        T* op = reinterpret_cast<T*>(p.out) + orow + n;
        if (ok) *reinterpret_cast<V8*>(op) = pk;
        if (PAIR && p.out_lo) {
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = v[e] - Tag::to_f32(pk[e]);
            constexpr bool CL = Tag::dtype == MVE_BF16;
            const u32x2 pl = {mve_lo8_pack4<CL>(r[0], r[1], r[2], r[3]), mve_lo8_pack4<CL>(r[4], r[5], r[6], r[7])};
            if (ok) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(p.out_lo) + orow + n) = pl;
        }
    }
}

// The fused epilogue on 8 consecutive output columns (n % 8 == 0) of row m: bias, per-image row vector (time embedding),
// GEGLU, residual, output scale, storage-dtype (or fp32) store.  Shared by the GEMM kernels and the split-K reducer.
// res_done: the caller's accumulators started from the residual (pair launches of the 128-row kernel): nothing to read or add here.
template <class Tag>
__device__ __forceinline__ void gemm_epilogue_store(const GemmParams& p, int m, int n, float (&v)[8], bool res_done = false) {
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    typedef T T4 __attribute__((ext_vector_type(4)));
    if (p.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if (p.rowvec) {
        const float* rv = p.rowvec + (size_t)(m / p.rows_per_vec) * p.ldrv + n;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(rv);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(rv + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if (p.geglu) {
        T4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = Tag::from_f32(v[2 * e] * gelu_erf(v[2 * e + 1]));
        *reinterpret_cast<T4*>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldc + (n >> 1)) = pk;
        return;
    }
    V8 rr = {};
    if (res_done) { gemm_epilogue_tail<Tag, true, true>(p, m, n, v, rr); return; }
    if (p.residual) rr = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(p.residual) + (size_t)m * p.ldr + n);
    gemm_epilogue_tail<Tag>(p, m, n, v, rr);
}

// In-kernel slice reduction (GemmParams::sk_sync).  Called by all threads of a block (PNTH threads, PBM x BN2 tile) after its fp32 partial tile has been
// written with write-through (sc1) stores (big_tile_epilogue's `partial` path; the 128-row kernel's epilogue).  Arrival: every wave drains its vmcnt, then one lane counts the block in
// (relaxed agent-scope atomic) and polls, relaxed, until the tile's S slices are all there; the partial tiles are read with sc1 loads -- correct
// for any placement of the slices on XCDs / CUs, without a release or acquire fence (MI355X_MICROARCH.md, inter-workgroup visibility).  Every block then folds rows [PBM s / S, PBM (s + 1) / S) of the tile over the slices in slice order 0 .. S - 1 starting from 0.0f and hands the
// sums to gemm_epilogue_store: exactly k_splitk_reduce's arithmetic, so the result is bit-identical to partials + reducer launch on any tile
// width and for any arrival order.  Waiting on siblings cannot deadlock: the launchers (gemm.hip: launch_red, launch_v) fold inside the launch only
// where the whole grid is resident at once (at most one ping-pong block, or two 128-row blocks, per CU), so a waiting block never holds the slot
// a sibling needs; a block only waits AFTER its own K loop.  Departure: the block that leaves last zeroes the counter for the next launch.
template <class Tag, int BN2, int PBM, int PNTH>
__device__ __forceinline__ void gemm_reduce_slices(const GemmParams& p, unsigned tile, int kslice, int S, int m0, int n0, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its write-through partial stores have reached memory
    __syncthreads();
    int* cnt = p.sk_sync + tile;
    if (tid == 0) {
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (relaxed polling: an acquire load invalidates the CU's L1 on every poll; nothing below needs an acquire -- the partial tiles are read sc1)
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    constexpr int CH = BN2 / 8;
    const int r_lo = PBM * kslice / S, r_hi = PBM * (kslice + 1) / S;
    const unsigned slice = (unsigned)((size_t)p.M * p.N * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, (int)0xFFFFFFF0u, 0x00020000);
    auto ld = [&](unsigned off) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);      // aux 16 = sc1: served past the L1, which may hold an older launch's lines
        return f32x4{__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
    };
    for (int task = tid; task < (r_hi - r_lo) * CH; task += PNTH) {
        const int r = task / CH, ch = task - r * CH;
        const int m = m0 + r_lo + r, n = n0 + ch * 8;
        if (m >= p.M || n >= p.N) continue;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned base = (unsigned)(((size_t)m * p.N + n) * 4);
        int s = 0;
        for (; s + 4 <= S; s += 4) {       // four slices requested before the first one is added; the additions stay in slice order
            f32x4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = ld(base + (unsigned)(s + u) * slice); b[u] = ld(base + (unsigned)(s + u) * slice + 16); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += a[u][e]; v[4 + e] += b[u][e]; }
        }
        for (; s < S; ++s) {
            const f32x4 a = ld(base + (unsigned)s * slice), b = ld(base + (unsigned)s * slice + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
        }
        gemm_epilogue_store<Tag>(p, m, n, v);
    }
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 2 * S - 1) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


}  // namespace
