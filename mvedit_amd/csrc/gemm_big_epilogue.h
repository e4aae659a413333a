// Epilogue of the 256-row GEMM / conv tiles (gemm_big.hip: 256 x {320,256} x 64 two-stage kernel; gemm_pp.hip: the ping-pong main loop
// over the same tile).  A wave (wm, wn) of the 2 x 4 arrangement holds acc[j][i] = 16 x 16 fragment (i along M, j along N) of its
// 128 x BN2/4 tile in the swapped-operand layout (a lane owns 4 consecutive n of one row).  `smem` is the kernel's whole dynamic LDS
// allocation: every wave must have finished reading it (and all LDS-DMA into it must have landed) before the call.
#pragma once
#include <type_traits>

#include "gemm_shared.h"
#include "ln_core.h"

namespace {

// Workgroup barrier for LDS hand-over only: LDS traffic of this wave has retired (lgkmcnt), global stores may still be in flight.
// __syncthreads() carries a release fence: hipcc drains vmcnt to 0 in front of it, so every pass of the epilogue waited for the write
// acknowledgements of its own 40 KiB of stores (~18 k cycles per tile for four passes, tools/pp_profile.py) although nothing after the
// barrier depends on them.
__device__ __forceinline__ void big_lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// compiler-level ordering fence: nothing is moved across it (loads placed behind it stay behind the LDS writes in front of it)
__device__ __forceinline__ void big_lds_fence() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// WAVES_N: waves along N (4: 2 x 4 arrangement, wave tile 128 x BN2/4;  2: 4 x 2 arrangement, wave tile 64 x BN2/2)
// PAIR: the kernel instantiation that serves the executor's residual_pair mode (GemmParams::residual_lo / out_lo).  A separate instantiation, not a
// run-time branch: with both fast paths in one function hipcc's register allocation of the ordinary one degrades from 1 to ~90 spilled registers.
// LNF (PAIR, BN2 = 320, N = 320 only): the fast path also writes LayerNorm(row) to GemmParams::ln_out -- see the LN phase below.
template <class Tag, int BN2, int WAVES_N = 4, bool PAIR = false, bool LNF = false>
__device__ __forceinline__ void big_tile_epilogue(const GemmParams& p, f32x4 (&acc)[BN2 / WAVES_N / 16][256 / (8 / WAVES_N) / 16], unsigned char* smem,
                                                  int m0, int n0, int kslice, int tid, int lane, int wm, int wn) {
    typedef typename Tag::V8 V8;
    typedef typename Tag::T T;
    constexpr int BM2 = 256, NTH = 512, WTM = 256 / (8 / WAVES_N), MF = WTM / 16;
    constexpr int WTN = BN2 / WAVES_N, NF = WTN / 16;
    constexpr int PPW = WTM / 64;                       // 64-row passes per wave row
    constexpr int CS_LD = BN2 + 4;
    // ---- GEGLU epilogue: out[m][i] = (v[2i] + b[2i]) * gelu(v[2i+1] + b[2i+1]).  A lane owns 4 consecutive n = two
    // (value, gate) pairs of one row, so the activation is evaluated on the accumulators; only the 16-bit results (half the
    // columns) pass through LDS, in ONE 256-row pass, for 16-byte row-segment stores.  Same arithmetic as gemm_epilogue_store.
    if (!PAIR && p.geglu && p.splitk <= 1) {
        constexpr int HS_LD = BN2 / 2 + 8;              // 168 elements: conflict-free 4-byte writes (row stride 84 words)
        T* Hs = reinterpret_cast<T*>(smem);
        typedef T T2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int c = wn * WTN + j * 16 + (lane >> 4) * 4;
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n0 + c < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const f32x4 v = acc[j][i];
                T2 o;
                o[0] = Tag::from_f32((v[0] + b4[0]) * gelu_erf(v[1] + b4[1]));
                o[1] = Tag::from_f32((v[2] + b4[2]) * gelu_erf(v[3] + b4[3]));
                *reinterpret_cast<T2*>(Hs + (wm * WTM + i * 16 + (lane & 15)) * HS_LD + (c >> 1)) = o;
            }
        }
        big_lds_barrier();
        constexpr int HCH = BN2 / 16;                   // 20 chunks of 8 output columns per row
        for (int task = tid; task < BM2 * HCH; task += NTH) {
            const int r = task / HCH, ch = task - r * HCH;
            const int m = m0 + r, n = n0 + ch * 16;     // n: first of the 16 input columns behind this output chunk
            if (m >= p.M || n >= p.N) continue;
            *reinterpret_cast<V8*>(reinterpret_cast<T*>(p.out) + (size_t)m * p.ldc + (n >> 1)) = *reinterpret_cast<const V8*>(Hs + r * HS_LD + ch * 8);
        }
        return;
    }

    // ---- epilogue: four 64-row passes through an fp32 LDS tile, 16-byte row-segment stores ----------------------------
    // A thread owns ONE 8-column chunk (ch) for the whole epilogue and walks rows r0, r0 + RPI, ... of each pass: bias is loaded
    // once, the per-image row vector once per pass, and the residual chunk of (pass + 1, k) is requested as soon as the registers of
    // (pass, k) are free -- a full pass before its use.  On gfx9 a wave's vector loads and stores retire in issue order, so a load
    // issued behind a store cannot return before that store is acknowledged (~2 k cycles): with the loads of a chunk issued right
    // before its arithmetic every chunk waited out the previous chunk's store (60-70 k cycles per tile with bias + residual against
    // 58 k for the whole K = 1280 loop, tools/pp_profile.py).
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CHUNKS = BN2 / 8;
    constexpr int RPI = NTH / CHUNKS;                   // rows per iteration: 12 (BN2 = 320, 480 of 512 threads busy) or 16
    constexpr int NIT = (64 + RPI - 1) / RPI;           // 6 | 4
    constexpr int NPASS = BM2 / 64;
    // Every lane runs every chunk on clamped indices and only the store is predicated: a divergent region around a chunk makes hipcc
    // copy the prefetched registers at its join, which needs their loads complete -- vmcnt(0) behind the chunk's own store again.
    const bool partial = p.splitk > 1;
    const int ch = tid % CHUNKS, r0 = tid / CHUNKS;
    const bool active = r0 < RPI;
    const int r0c = active ? r0 : RPI - 1;
    const int n = n0 + ch * 8;
    const int nc = n < p.N ? n : p.N - 8;
    const bool col_ok = active && n < p.N;
    const bool rv_pass = p.rowvec && p.rows_per_vec % 64 == 0;     // a 64-row pass lies inside one image (m0 is a multiple of 256)
    f32x4 b0 = {}, b1 = {}, rv0 = {}, rv1 = {};
    V8 rr[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) rr[k] = V8{};
    auto row_c = [&](int pass, int k) {                 // clamped global row of (pass, k)
        const int r = r0c + RPI * k;
        const int m = m0 + pass * 64 + (r < 64 ? r : 63);
        return m < p.M ? m : p.M - 1;
    };
    auto load_rv = [&](int m, f32x4& a, f32x4& b) {
        const float* rv = p.rowvec + (size_t)(m / p.rows_per_vec) * p.ldrv + nc;
        a = *reinterpret_cast<const f32x4*>(rv);
        b = *reinterpret_cast<const f32x4*>(rv + 4);
    };
    if (!partial) {
        if (p.bias) {
            b0 = *reinterpret_cast<const f32x4*>(p.bias + nc);
            b1 = *reinterpret_cast<const f32x4*>(p.bias + nc + 4);
        }
        if (rv_pass) load_rv(m0 < p.M ? m0 : p.M - 1, rv0, rv1);
        if (!PAIR && p.residual) {         // (PAIR: the residual is inside the accumulators already, or -- split K -- the reducer's business)
#pragma unroll
            for (int k = 0; k < NIT; ++k) rr[k] = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(p.residual) + (size_t)row_c(0, k) * p.ldr + nc);
        }
    }
    // ---- fast path: the configuration every UNet / VAE / ControlNet launch of an interior tile has (bias, unit output scale, 16-bit
    // output, residual before the scale, row vector constant over a pass).  Written without per-chunk tests: left to if-convert the
    // generic code below, hipcc evaluates both sides of `if (p.bias)` / `if (scale != 1)` and selects (8 v_cndmask each) and redoes the
    // 64-bit row address products per chunk; the epilogue is VALU-bound (2 waves per SIMD, ~70 VALU per chunk there, ~30 here).
    // Same operations in the same order as gemm_epilogue_tail, so the bits are the same.
    const bool fast = !partial && p.bias && p.out_scale == 1.0f && !p.res_after_scale && !p.out_f32 && (!p.rowvec || rv_pass) &&
                      m0 + BM2 <= p.M && n0 + BN2 <= p.N && p.orow_extra == 0;      // (grouped output rows: the generic path below addresses per row)
    // ---- residual_pair mode (PAIR).  The residual pair is NOT read here: the PAIR kernel starts its accumulators from it (k_gemm_pp), so this
    // epilogue has no residual to prefetch next to the 160 accumulators (reading both halves a pass ahead needed 48 registers in flight and
    // spilled into the chunk loop).  The result leaves as hi = round16(v) and the 8-bit low half lo8(v - hi) (common.h).
    // Round 5: this is the ordinary fast path below with RES = false plus the 8-byte lo8 store.  Round 4's own path (32-row passes from both wave
    // rows, buffer-resource stores) left ~1e-5 of the elements of large launches holding stale LDS contents -- a race that only B >= 16 forwards
    // reach and no round-4 test ran (profiles/r05_debug_pair_ops*.log: per-element errors of +-512 at tile rows 70-107, columns = 32..39 mod 64;
    // the generic path below was correct on the same launches) -- and is gone.
    if (fast && (!PAIR || (p.out_lo && !(p.dbg & 2)))) {          // (MVE_PP_DBG bit 1: the pair launches take the generic path below -- A/B aid)
        auto run_fast = [&](auto rv_c, auto res_c) {
#pragma clang fp contract(off)
            constexpr bool RV = decltype(rv_c)::value, RES = decltype(res_c)::value;
            T* outb = reinterpret_cast<T*>(p.out) + (size_t)(m0 + r0c) * p.ldc + n;
            unsigned char* outl = reinterpret_cast<unsigned char*>(p.out_lo) + (size_t)(m0 + r0c) * p.ldc + n;      // (PAIR only)
            const T* resb = reinterpret_cast<const T*>(p.residual) + (size_t)(m0 + r0c) * p.ldr + n;
            // row offset (relative to r0c) of chunk k: rows past the pass (BN2 = 320: k = 5 for r0 >= 4) re-read row k - 1 and store nothing
            auto rel_row = [&](int k) { return (r0c + RPI * k < 64) ? RPI * k : RPI * (k - 1); };
            const f32x4 fb0 = *reinterpret_cast<const f32x4*>(p.bias + n), fb1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
            f32x4 frv0 = {}, frv1 = {};
            V8 fr[NIT];
            if constexpr (RV) load_rv(m0, frv0, frv1);
            if constexpr (RES) {
#pragma unroll
                for (int k = 0; k < NIT; ++k) fr[k] = *reinterpret_cast<const V8*>(resb + (size_t)rel_row(k) * p.ldr);
            }
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                if (wm == pass / PPW) {
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) {
                            const int i = (pass % PPW) * 4 + i4;
                            const int r = i4 * 16 + (lane & 15);
                            const int c = wn * WTN + j * 16 + (lane >> 4) * 4;
                            *reinterpret_cast<f32x4*>(Cs + r * CS_LD + c) = acc[j][i];
                        }
                }
                big_lds_barrier();
                const f32x4 crv0 = frv0, crv1 = frv1;
                if constexpr (RV) { if (pass + 1 < NPASS) load_rv(m0 + (pass + 1) * 64, frv0, frv1); }
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int rr_ = rel_row(k);
                    const bool ok = active && r0c + RPI * k < 64;
                    const float* cs = Cs + (r0c + rr_) * CS_LD + ch * 8;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(cs), hi = *reinterpret_cast<const f32x4*>(cs + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = lo[e] + fb0[e]; v[4 + e] = hi[e] + fb1[e]; }
                    if constexpr (RV) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += crv0[e]; v[4 + e] += crv1[e]; }
                    }
                    if constexpr (RES) {
                        const V8 res = fr[k];
                        if (pass + 1 < NPASS) fr[k] = *reinterpret_cast<const V8*>(resb + (size_t)((pass + 1) * 64 + rr_) * p.ldr);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += Tag::to_f32(res[e]);
                    }
                    V8 pk;
                    if constexpr (PAIR) {
                        const u32x2 pl = mve_pair_split8<Tag>(v, pk);
                        if (ok) {
                            *reinterpret_cast<V8*>(outb + (size_t)(pass * 64 + rr_) * p.ldc) = pk;
                            *reinterpret_cast<u32x2*>(outl + (size_t)(pass * 64 + rr_) * p.ldc) = pl;
                        }
                        if constexpr (LNF) {
                            // the value the LayerNorm kernel would read back: hi + lo8 (exact in fp32) -- parked where this lane found its accumulators
                            float x8[8];
                            mve_pair_load8<Tag>(pk, pl, x8);
                            if (ok) {
                                float* xs = Cs + (r0c + rr_) * CS_LD + ch * 8;
                                *reinterpret_cast<f32x4*>(xs) = f32x4{x8[0], x8[1], x8[2], x8[3]};
                                *reinterpret_cast<f32x4*>(xs + 4) = f32x4{x8[4], x8[5], x8[6], x8[7]};
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) pk[e] = Tag::from_f32(v[e]);
                        if (ok) *reinterpret_cast<V8*>(outb + (size_t)(pass * 64 + rr_) * p.ldc) = pk;
                    }
                }
                if constexpr (LNF) {
                    // LN phase of the pass: 64 rows of 320 channels sit in the staging tile as the LayerNorm kernel would load them; wave w normalises
                    // rows 8 w .. 8 w + 7 with that kernel's row arithmetic (ln_core.h: lane l < 40 holds chunk l) and writes them to ln_out
                    big_lds_barrier();
                    const int wv = tid >> 6;
                    const bool has = lane < BN2 / 8;
                    const int lc = has ? lane : 0;
                    T* lnb = reinterpret_cast<T*>(p.ln_out) + (size_t)(m0 + pass * 64 + wv * 8) * p.ld_ln + lc * 8;
                    constexpr int LNR = 1;
                    // LNR rows at a time: their reduction trees are independent, so one row's shuffle latency (ds_bpermute) hides behind the others'
                    // (one row at a time this phase cost as much as the LayerNorm kernel it replaces: profiles/r06_ln_epilogue_ab.log)
                    const float* g8 = p.ln_gamma + lc * 8;      // (read per row from L1: sixteen more live registers next to the accumulators of the later passes spill)
                    const float* b8 = p.ln_beta + lc * 8;
#pragma unroll
                    for (int r4 = 0; r4 < 8; r4 += LNR) {
                        float x[LNR][8], sm[LNR], q[LNR];
#pragma unroll
                        for (int r = 0; r < LNR; ++r) {
                            const float* xs = Cs + (wv * 8 + r4 + r) * CS_LD + lc * 8;
                            const f32x4 a = *reinterpret_cast<const f32x4*>(xs), b = *reinterpret_cast<const f32x4*>(xs + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { x[r][e] = a[e]; x[r][4 + e] = b[e]; }
                            sm[r] = has ? mve_ln_sum8(x[r], 0.f) : 0.f;
                        }
                        mve_ln_wave_sum_n<LNR>(sm);
#pragma unroll
                        for (int r = 0; r < LNR; ++r) { sm[r] = sm[r] / (float)BN2; q[r] = has ? mve_ln_sq8(x[r], sm[r], 0.f) : 0.f; }
                        mve_ln_wave_sum_n<LNR>(q);
#pragma unroll
                        for (int r = 0; r < LNR; ++r) {
                            const float rstd = rsqrtf(q[r] / (float)BN2 + p.ln_eps);
                            const V8 y = mve_ln_out8<Tag>(x[r], sm[r], rstd, g8, b8);
                            if (has) *reinterpret_cast<V8*>(lnb + (size_t)(r4 + r) * p.ld_ln) = y;
                        }
                    }
                }
                big_lds_barrier();
            }
        };
        if constexpr (PAIR) {          // (the residual is inside the accumulators: RES = false)
            if (p.rowvec) run_fast(std::true_type(), std::false_type()); else run_fast(std::false_type(), std::false_type());
        } else {
            if (p.rowvec) { if (p.residual) run_fast(std::true_type(), std::true_type()); else run_fast(std::true_type(), std::false_type()); }
            else { if (p.residual) run_fast(std::false_type(), std::true_type()); else run_fast(std::false_type(), std::false_type()); }
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t part_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, (int)0xFFFFFFF0u, 0x00020000);      // (sk_sync launches only)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (wm == pass / PPW) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const int i = (pass % PPW) * 4 + i4;
                    const int r = i4 * 16 + (lane & 15);
                    const int c = wn * WTN + j * 16 + (lane >> 4) * 4;
                    *reinterpret_cast<f32x4*>(Cs + r * CS_LD + c) = acc[j][i];
                }
        }
        big_lds_barrier();
        const f32x4 crv0 = rv0, crv1 = rv1;
        if (!partial && rv_pass && pass + 1 < NPASS) {
            const int m = m0 + (pass + 1) * 64;
            load_rv(m < p.M ? m : p.M - 1, rv0, rv1);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int r = r0c + RPI * k;
            const int rcl = r < 64 ? r : 63;
            const int m = m0 + pass * 64 + rcl;
            const int mc = m < p.M ? m : p.M - 1;
            const bool ok = col_ok && r < 64 && m < p.M;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + rcl * CS_LD + ch * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + rcl * CS_LD + ch * 8 + 4);
            const V8 res = rr[k];
            if (!PAIR && !partial && p.residual && pass + 1 < NPASS)
                rr[k] = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(p.residual) + (size_t)row_c(pass + 1, k) * p.ldr + nc);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
            if (partial) {
                if (p.sk_sync) {
                    // slices folded inside the launch (pp_reduce_slices): the partial tile leaves WRITE-THROUGH (sc1, 16-byte buffer stores) -- visible to
                    // the sibling blocks on any XCD once this wave's vmcnt has drained, with no L2 write-back fence (a release fence per block
                    // writes back the XCD's whole L2: measured +100 us on a 256-block launch, profiles/r06_ab_red_v1_fences.log)
                    const unsigned off = (unsigned)((((size_t)kslice * p.M + mc) * p.N + nc) * 4);
                    if (ok) {
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, part_rsrc, off, 0, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, part_rsrc, off + 16, 0, 16);
                    }
                    continue;
                }
                float* pp = p.partial + ((size_t)kslice * p.M + mc) * p.N + nc;
                if (ok) {
                    *reinterpret_cast<f32x4*>(pp) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(pp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
                continue;
            }
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
            }
            if (p.rowvec) {
                f32x4 a = crv0, b = crv1;
                if (!rv_pass) load_rv(mc, a, b);         // general case: one row vector per row
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
            }
            gemm_epilogue_tail<Tag, PAIR, PAIR>(p, mc, nc, v, res, ok);
        }
        big_lds_barrier();
    }
}

}  // namespace
