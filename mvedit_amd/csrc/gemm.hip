// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (fp16 or bf16 storage, fp32 accumulate).
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
//
// A is either a dense row-major matrix (Linear / 1x1 conv over NHWC activations) or the im2col
// view of an NHWC activation tensor gathered on the fly (3x3 conv, pad 1; stride 1 or 2; optional
// nearest 2x upsample of the input; optional channel-concat of two inputs).  W is [N][K] row-major
// ("B^T"), which is PyTorch's Linear layout and, for convs, [Cout][kh][kw][Cin] or the channel-chunk-major
// [Cout][Cin/64][kh][kw][64] (MVE_CONV_W_CHUNK64): with the latter the nine taps of one 64-channel slab are
// consecutive K tiles, so the 3x3 re-reads of an input pixel happen within nine tiles and stay in L1/L2
// instead of being a full pass over the tile apart (rocprof FETCH_SIZE, profiles/r01_*).
//
// Work decomposition (all wave64):
//   block tile 128 x BN (BN = 128 | 160 | 64) x 64, 256 threads = 4 waves in a 2x2 grid;
//   each wave owns a 64 x BN/2 sub-tile = 4 x (BN/32) fragments of v_mfma_f32_16x16x32_{f16,bf16};
//   the MFMA is issued "swapped" (W fragment as the A operand, activation fragment as B), so every
//   lane ends up with 4 consecutive n for one m and moves whole float4s into an fp32 LDS staging tile (two
//   64-row passes); bias / time-embedding / residual / GEGLU are applied in fp32 on the way out and the global
//   stores are full 16-byte row segments;
//   LDS tiles are [rows][64] 16-bit with a 16-byte-chunk XOR swizzle chunk ^= (row>>1)&7 that makes
//   the ds_read_b128 fragment reads conflict free;
//   staging is LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, lane-linear destination): the
//   swizzle is applied to the per-lane SOURCE chunk, zero padding (conv halo, K tail) is sourced from a
//   16-byte zero page; tile k+1 is in flight while tile k feeds the MFMAs, one barrier per K tile;
//   blockIdx -> tile uses the XCD-aware bijective remap so that tiles sharing an activation row panel
//   run on one XCD (one L2).
// (A register-staged variant -- global -> VGPR -> ds_write -- measured 4 % slower end to end, profiles/r01_bench_v0.log vs
// r01_bench_v1_dma.log, and was removed.)
//
// Replaces (behaviourally) the cuDNN/cuBLAS calls behind diffusers' ResnetBlock2D / Attention /
// FeedForward as driven by lib/models/architecture/diffusers.py:57-164 of the reference.
#include "common.h"

#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "gemm_shared.h"

namespace {

constexpr int BM = 128;

}  // namespace
// defined in gemm_big.hip (256 x 320 tiles, bit-identical results)
long long mve_gemm_big_blocks(int M, int N, int splitk);
int mve_gemm_big_launch(int dtype, int mode, const void* params, void* stream);
int mve_gemm_pp_launch(int dtype, int mode, const void* params, void* stream);      // gemm_pp.hip; 1 = not eligible
void mve_gemm_pp_old_swizzle(int on);
bool mve_gemm_pp_ln_fused();
int mve_gemm_pp_ln_fuse_tune(int on);
extern "C" int mve_layernorm_pair(int, const void*, int, void*, int, int, int, const float*, const float*, float, const void*, void*);
namespace {

// NST: stages of the LDS ring.  2: the two-stage loop of rounds 1-5 (72 KiB at BN = 160: two blocks per CU; tile k + 1 in flight under the MFMAs of
// tile k, drained at every barrier).  4 (round 6, k_gemm_deep; 144 KiB: one block per CU): launches of at most a block or two per CU -- the K-sliced
// GEMMs / convs of the deep UNet levels when a rank holds few images -- are a chain of K-tile round trips (measured 1.0-1.5 us per 64-wide K tile against
// ~0.3 us of MFMA work: time_embedding.linear_2, 20 tiles, 26.6 us; the 8 x 8 conv at 8 images, 22 tiles per slice, 40 us).  The ring keeps THREE tiles in
// flight and never drains: LDS-DMA pieces issued from inline asm (M0 owned by the loop), one `s_waitcnt vmcnt(2 x pieces)` + `s_barrier` per tile.
// Same tile, MFMA order and epilogue: bit-identical results.
// BM (round 6): rows of the tile, 128 or 64.  64 x 64 tiles serve launches so small that even 128 x 64 tiles leave CUs with a single block (launch_v):
// the waves stay 2 x 2, a wave owns 32 rows, the epilogue is ONE 64-row pass written by both wave rows.  Same K order and epilogue arithmetic: same bits.
template <class Tag, int BN, int MODE, int NST, int BM = 128>   // MODE 0: dense A, 1: conv3x3 gather
__device__ __forceinline__ void gemm_body(const GemmParams& p, unsigned char* smem) {
    static_assert(BM == 128 || BM == 64, "tile rows");
    constexpr int WN = BN / 2;          // wave sub-tile width
    constexpr int NF = WN / 16;         // W fragments per wave (4 / 5 / 2)
    constexpr int WM = BM / 2;          // rows of a wave's sub-tile (64 / 32)
    constexpr int MF = WM / 16;         // activation fragments per wave (4 / 2)
    constexpr int ROWS_PER_PASS = NT / 8;               // 32 rows per load pass
    constexpr int A_PASSES = BM / ROWS_PER_PASS;        // 4
    constexpr int B_PASSES = BN / ROWS_PER_PASS;        // 4 / 5 / 2
    constexpr int A_STAGE = BM * ROW_BYTES;             // 16 KB
    constexpr int B_STAGE = BN * ROW_BYTES;
    constexpr int STAGE = A_STAGE + B_STAGE;
    constexpr int CS_LD = BN + 4;                       // fp32 epilogue staging row stride (floats)
    typedef typename Tag::V8 V8;
    typedef typename Tag::T T;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // split-K: the K slices of one output tile are adjacent logical blocks (same XCD, concurrently resident)
    const int S = p.splitk > 1 ? p.splitk : 1;
    const unsigned lin = mve_xcd_remap(blockIdx.x, (unsigned)(tiles_m * tiles_n * S));
    int kslice, tm, tn;
    if (p.w_major) {          // weight strip major (GemmParams::w_major): the row panels of one (column tile, K slice) strip are consecutive blocks
        tm = (int)(lin % (unsigned)tiles_m);
        const unsigned rest = lin / (unsigned)tiles_m;
        kslice = (int)(rest % (unsigned)S);
        tn = (int)(rest / (unsigned)S);
    } else {
        kslice = lin % S;
        const unsigned tile = lin / S;
        tm = tile / tiles_n; tn = tile % tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread load coordinates ------------------------------------------------------
    const int lr = tid >> 3;           // row within a 32-row pass
    // the LDS-DMA writes LDS lane-linearly, so the lane at physical chunk (tid&7) must FETCH the logical chunk that the
    // swizzle maps there; (row>>1)&7 == (lr>>1)&7 for every pass because passes advance by 32 rows.
    const int lc = (tid & 7) ^ ((lr >> 1) & 7);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.W);

    const T* a_row[A_PASSES];
    int cb[A_PASSES], cy[A_PASSES], cx[A_PASSES];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            int m = m0 + lr + j * ROWS_PER_PASS;
            m = m < p.M ? m : p.M - 1;
            a_row[j] = reinterpret_cast<const T*>(p.A) + (size_t)m * p.lda;
        }
    } else {
        const int hw = p.g.Ho * p.g.Wo;
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            int m = m0 + lr + j * ROWS_PER_PASS;
            m = m < p.M ? m : p.M - 1;
            const int b = m / hw, r = m - b * hw;
            const int y = r / p.g.Wo;
            cb[j] = b;
            cy[j] = y * p.g.stride - p.g.pad;
            cx[j] = (r - y * p.g.Wo) * p.g.stride - p.g.pad_x;
        }
    }
    const T* w_row[B_PASSES];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
        int n = n0 + lr + j * ROWS_PER_PASS;
        n = n < p.N ? n : p.N - 1;
        w_row[j] = Wp + (size_t)n * p.ldw;
    }

    // conv, tap-major K order: running (tap, cin) of this thread's chunk
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt_begin = (int)((long long)nk_all * kslice / S), kt_end = (int)((long long)nk_all * (kslice + 1) / S);
    const int Ctot = p.g.C1 + p.g.C2;
    int tap = 0, cin = 0;
    if constexpr (MODE == 1) {
        if (!p.g.chunk64) {
            const int k0 = kt_begin * BK + lc * 8;
            tap = k0 / Ctot;
            cin = k0 - tap * Ctot;
        }
    }
    const T* zero = reinterpret_cast<const T*>(g_zero_page);

    // source pointers of this thread's chunks of K tile kt (nullptr = zero fill)
    auto a_src = [&](int kt, int j, bool kin) -> const T* {
        if constexpr (MODE == 0) {
            return kin ? a_row[j] + kt * BK + lc * 8 : nullptr;
        } else {
            int t_ = tap, c_ = cin;
            if (p.g.chunk64) { t_ = kt % 9; c_ = (kt / 9) * 64 + lc * 8; }
            return conv_src<T>(p, cb[j], cy[j], cx[j], t_, c_, kin);
        }
    };
    auto advance = [&]() {
        if constexpr (MODE == 1) {
            if (!p.g.chunk64) {
                cin += BK;
                while (cin >= Ctot) { cin -= Ctot; ++tap; }
            }
        }
    };

    // ---- staging ---------------------------------------------------------------------------------
    // fast conv addressing for the slab-major K order without upsample (see gemm_big.hip): uniform tap / slab arithmetic,
    // one multiply-add per row, the halo test is a bit of a per-row mask computed once
    const bool fast = MODE == 1 && p.g.chunk64;
    const int gC1 = p.g.C1, gC3 = p.g.C3, g_nkm = p.g.nk_main;
    const int dC21 = p.g.C2 - p.g.C1, dC43 = p.g.C4 - p.g.C3;
    const long long gA1 = (long long)p.A, gA3 = (long long)p.A3;
    const long long dA21 = (long long)p.A2 - (long long)p.A, dA43 = (long long)p.A4 - (long long)p.A3;
    int pix[A_PASSES];
    unsigned vmask[A_PASSES];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            pix[j] = (cb[j] * p.g.Hs + (cy[j] >> p.g.ups)) * p.g.Ws + (cx[j] >> p.g.ups);     // upsample: see gemm_big.hip
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yi = cy[j] + t / 3, xi = cx[j] + t % 3;
                if (yi >= 0 && yi < p.g.Hv && xi >= 0 && xi < p.g.Wv) mk |= 1u << t;
            }
            if (p.g.ups) mk |= ((unsigned)(cy[j] & 1) << 9) | ((unsigned)(cx[j] & 1) << 10);
            vmask[j] = mk;
        }
    }
    // one LDS-DMA piece: 64 lanes x 16 B, lane-linear from a wave-uniform LDS address.  Ring of more than two stages: issued from inline asm, so that
    // the compiler neither counts it nor drains it (the waits are the loop's own, below)
    const unsigned smem_lds = (unsigned)(size_t)smem;
    auto piece = [&](const T* src, unsigned char* dst) {
        if constexpr (NST > 2) {
            const unsigned d = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_lds + (unsigned)(dst - smem)));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(d) : "memory");
        } else {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
        }
    };
    auto dma_tile = [&](int kt, int stage) {   // global -> LDS directly, 1 KiB per wave instruction
#ifdef MVE_GEMM_LAB
        if ((p.dbg & 32) && kt > kt_begin + 2) return;        // timing-only ablation (tools/ab_gemm_lab.py): no LDS-DMA after the first tiles
#endif
        const bool kin = kt * BK + lc * 8 < p.K;
        unsigned char* As = smem + stage * STAGE;
        unsigned char* Bs = As + A_STAGE;
        if (MODE == 1 && fast) {
            // (sources as arithmetic on opaque scalars, as in gemm_pp.hip: left to select between kernarg FIELDS -- `second ? p.A2 : p.A` -- hipcc copies
            // the parameter block to scratch and indexes it, and every scratch load in the K loop waits vmcnt(0): the whole LDS-DMA queue)
            int t_ = kt % 9, c0 = (kt / 9) * 64;                       // uniform
            bool second = c0 >= gC1;
            long long srcb = gA1 + (second ? dA21 : 0ll);
            int cs = gC1 + (second ? dC21 : 0);
            int ch = (second ? c0 - gC1 : c0) + lc * 8;
            if (g_nkm > 0 && kt >= g_nkm) {                            // 1x1 shortcut part: centre tap of the shortcut sources
                t_ = 4;
                c0 = (kt - g_nkm) * 64;
                second = c0 >= gC3;
                srcb = gA3 + (second ? dA43 : 0ll);
                cs = gC3 + (second ? dC43 : 0);
                ch = (second ? c0 - gC3 : c0) + lc * 8;
            }
            const T* src = reinterpret_cast<const T*>(srcb);
            const int dy = t_ / 3, dx = t_ - dy * 3;
            const int toff = dy * p.g.Ws + dx;
#pragma unroll
            for (int j = 0; j < A_PASSES; ++j) {
                int to = toff;
                if (p.g.ups) to = (int)((((vmask[j] >> 9) & 1u) + dy) >> 1) * p.g.Ws + (int)((((vmask[j] >> 10) & 1u) + dx) >> 1);
                const unsigned off = (unsigned)((pix[j] + to) * cs + ch);
                const T* s = (kin && ((vmask[j] >> t_) & 1u)) ? src + off : zero;      // (kin: the ring issues tiles past the end of K, never read)
                piece(s, As + (j * ROWS_PER_PASS + wid * 8) * ROW_BYTES);
            }
        } else
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            const T* s = a_src(kt, j, kin);
            s = s ? s : zero;
            piece(s, As + (j * ROWS_PER_PASS + wid * 8) * ROW_BYTES);
        }
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) {
            const T* s = kin ? w_row[j] + kt * BK + lc * 8 : zero;
            piece(s, Bs + (j * ROWS_PER_PASS + wid * 8) * ROW_BYTES);
        }
        advance();
    };

    f32x4 acc[NF][MF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // residual_pair launches (a low half comes in or goes out): the accumulators START from the residual, hi + lo (exact in fp32), exactly as the
    // PAIR instantiation of k_gemm_pp does -- residual + sum_k a w, then + bias -- so that a pair launch rounds identically on every tile the
    // dispatcher may pick (batch / partition invariance in the executor's default mode).  A split-K slice starts from zero: the reducer adds it.
    const bool res_in_acc = (p.residual_lo || p.out_lo) && p.residual && p.splitk <= 1 && !p.res_after_scale;      // uniform
    if (res_in_acc) {
        typedef T T4 __attribute__((ext_vector_type(4)));
        const T* rh = reinterpret_cast<const T*>(p.residual);
        const unsigned char* rl = reinterpret_cast<const unsigned char*>(p.residual_lo);      // lo8: one byte per element (common.h)
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            int m = m0 + wm * WM + i * 16 + (lane & 15);
            m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                int n = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
                n = n + 4 <= p.N ? n : 0;                     // (a dead column group of the last tile: never stored)
                const size_t ro = (size_t)m * p.ldr + n;
                const T4 h = *reinterpret_cast<const T4*>(rh + ro);
                float l[4] = {0.f, 0.f, 0.f, 0.f};
                if (rl) mve_lo8_unpack4(*reinterpret_cast<const unsigned*>(rl + ro), l);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][i][e] = Tag::to_f32(h[e]) + l[e];
            }
        }
    }

    const int frow = lane & 15, fchunk = lane >> 4;
    auto mma_tile = [&](int stage) {
#ifdef MVE_GEMM_LAB
        if (p.dbg & 16) return;                               // timing-only ablation: no fragment reads, no MFMAs
#endif
        const unsigned char* As = smem + stage * STAGE;
        const unsigned char* Bs = As + A_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            V8 xf[MF], wf[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i)
                xf[i] = *reinterpret_cast<const V8*>(As + swz(wm * WM + i * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int j = 0; j < NF; ++j)
                wf[j] = *reinterpret_cast<const V8*>(Bs + swz(wn * WN + j * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int i = 0; i < MF; ++i) acc[j][i] = Tag::mfma16(wf[j], xf[i], acc[j][i]);
        }
    };
    if constexpr (NST > 2) {
        // Ring of NST stages, prefetch distance D = NST - 1 tiles, never drained.  Iteration i (tile kt = kt_begin + i, stage i % NST):
        //   wait until this wave's pieces of tile kt have landed -- D tiles are in flight, D - 1 may stay: vmcnt((D - 1) x pieces) -- and meet the
        //   other waves (RAW: their pieces of tile kt too;  WAR: every wave is done with the MFMAs of tile kt - 1, whose stage (i - 1) % NST =
        //   (i + D) % NST is the one refilled next);  issue tile kt + D;  MFMAs of tile kt.
        // Tiles past kt_end are issued all the same (their K columns lie past the slice, or past K: zero page) and never read: the count of pieces in
        // flight stays uniform, no tail code.  The compiler's own loads (res_in_acc above) are drained first so that they cannot sit in the queue.
        constexpr int D = NST - 1, PIECES = A_PASSES + B_PASSES;
        static_assert((D - 1) * PIECES <= 63, "vmcnt range");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned m0_keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(m0_keep));
#pragma unroll 1
        for (int t = 0; t < D; ++t) dma_tile(kt_begin + t, t);
        int st = 0;
#pragma unroll 1
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((D - 1) * PIECES) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const int st_new = st == 0 ? NST - 1 : st - 1;      // (i + D) % NST
            dma_tile(kt + D, st_new);
            mma_tile(st);
            st = st + 1 == NST ? 0 : st + 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // the pieces past the end have landed before the epilogue reuses the ring
        asm volatile("s_mov_b32 m0, %0" ::"s"(m0_keep));
        __builtin_amdgcn_sched_barrier(0);
    } else {
    dma_tile(kt_begin, 0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) {
            dma_tile(kt + 1, cur ^ 1);
        }
        mma_tile(cur);
        __syncthreads();   // the compiler drains the in-flight LDS-DMA (vmcnt(0)) ahead of this barrier
    }
    }

    {
#ifdef MVE_GEMM_LAB
        if (p.dbg & 64) return;                               // timing-only ablation: no epilogue
#endif
        // ---- epilogue: two 64-row passes through an fp32 LDS tile, 16-byte row-segment stores ----------------
        // (measured: storing 8 bytes per lane straight from the accumulators was 10-20 % slower on the K = 320 GEMMs)
        float* Cs = reinterpret_cast<float*>(smem);
        constexpr int CHUNKS = BN / 8;                 // 8-column chunks per row
        constexpr int TASKS = 64 * CHUNKS;
#pragma unroll 1
        for (int pass = 0; pass < BM / 64; ++pass) {
            if (BM == 64 || wm == pass) {          // (64-row tiles: one pass, both wave rows write their 32 rows)
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int i = 0; i < MF; ++i) {
                        const int r = (BM == 64 ? wm * WM : 0) + i * 16 + (lane & 15);
                        const int c = wn * WN + j * 16 + (lane >> 4) * 4;
                        *reinterpret_cast<f32x4*>(Cs + r * CS_LD + c) = acc[j][i];
                    }
            }
            __syncthreads();
            for (int task = tid; task < TASKS; task += NT) {
                const int r = task / CHUNKS, ch = task - r * CHUNKS;
                const int m = m0 + pass * 64 + r, n = n0 + ch * 8;
                if (m >= p.M || n >= p.N) continue;      // N is a multiple of 8: whole chunk in or out
                float v[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + r * CS_LD + ch * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + r * CS_LD + ch * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                if (p.splitk > 1 && p.sk_sync) {      // slices folded inside the launch (gemm_reduce_slices below): the partial tile leaves write-through
                    const unsigned off = (unsigned)((((size_t)kslice * p.M + m) * p.N + n) * 4);
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, (int)0xFFFFFFF0u, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rs, off, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, rs, off + 16, 0, 16);
                    continue;
                }
                if (p.splitk > 1) {      // raw fp32 partial tile of this K slice; the reducer applies the epilogue
                    float* pp = p.partial + ((size_t)kslice * p.M + m) * p.N + n;
                    *reinterpret_cast<f32x4*>(pp) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(pp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    continue;
                }
                gemm_epilogue_store<Tag>(p, m, n, v, res_in_acc);
            }
            __syncthreads();
        }
    }
    if (p.splitk > 1 && p.sk_sync) gemm_reduce_slices<Tag, BN, BM, NT>(p, (unsigned)(tm * tiles_n + tn), kslice, S, m0, n0, tid);      // (uniform branch)
}

template <int BN, int NST, int BMT = BM>
constexpr int gemm_smem_bytes() {
    return (NST * (BMT + BN) * ROW_BYTES > 64 * (BN + 4) * 4) ? NST * (BMT + BN) * ROW_BYTES : 64 * (BN + 4) * 4;
}

template <class Tag, int BN, int MODE>
__global__ __launch_bounds__(NT, 2) void k_gemm(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[gemm_smem_bytes<BN, 2>()];
    gemm_body<Tag, BN, MODE, 2>(p, smem);
}

template <class Tag, int BN, int MODE>
__global__ __launch_bounds__(NT, 2) void k_gemm64(const GemmParams p) {      // 64-row tiles (gemm_body's BM)
    __shared__ __attribute__((aligned(16))) unsigned char smem[gemm_smem_bytes<BN, 2, 64>()];
    gemm_body<Tag, BN, MODE, 2, 64>(p, smem);
}

constexpr int DEEP_NST = 4;
template <class Tag, int BN, int MODE>
__global__ __launch_bounds__(NT, 1) void k_gemm_deep(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    gemm_body<Tag, BN, MODE, DEEP_NST>(p, dyn_smem);
}

// split-K reducer: out = epilogue( sum_s partial[s] ), 8 columns per thread
template <class Tag>
__global__ __launch_bounds__(256) void k_splitk_reduce(const GemmParams p) {
    const size_t chunks_n = p.N / 8;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)p.M * chunks_n) return;
    const int m = (int)(i / chunks_n), n = (int)(i - (size_t)m * chunks_n) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splitk; ++s) {        // fixed order: deterministic
        const float* pp = p.partial + ((size_t)s * p.M + m) * p.N + n;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(pp), hi = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += lo[e]; v[4 + e] += hi[e]; }
    }
    gemm_epilogue_store<Tag>(p, m, n, v);
}

// Split K at the deep UNet levels, where one image contributes only a few output tiles (8x8 / 16x16 latents) but K is
// 9*1280..9*2560.  The RULE below is a function of (rows per image, N, K) only -- never of the batch.  What a launch actually runs with is the
// rule's count only while the launch is small: launch_gemm (below) runs ONE accumulation chain where the un-split launch fills the chip and
// `ceil(256 / tiles)` slices where the rule would over-fill it, so the effective count falls with the rows of the launch
// (mve_gemm_effective_splitk; 16 x 16 level: 4 slices up to 16 images, 2 at 32, 1 from 64; 8 x 8 level: 8 / 4 / 2 / 1 slices at <= 32 / 64 / 128 /
// 256 images).  A view therefore gets bit-identical results alone, in a chunk or on another rank AS LONG AS those launches take the same decision
// (all small batches do); MVE_GEMM_STRICT_SPLITK=1 / mve_gemm_tune bit 30 / mvedit_amd.parallel.set_partition_invariant() make every launch
// round as the rule's slices, at any batch (tests/test_abi.py::test_effective_splitk_by_batch).
template <class Tag>
int splitk_reduce_launch(const GemmParams& p, hipStream_t s) {
    if (p.splitk > 1 && !p.sk_sync) {
        k_splitk_reduce<Tag><<<mve_cdiv((size_t)p.M * (p.N / 8), 256), 256, 0, s>>>(p);
        MVE_LAUNCH_CHECK();
    }
    return MVE_OK;
}

int g_splitk_policy = -1;      // MVE_GEMM_SPLITK: 1 (default) = the rule below; 0 = never split (A/B: what the slices cost at a given batch)
int choose_splitk(int rows_per_image, int N, int K) {
    if (g_splitk_policy < 0) {
        const char* e = getenv("MVE_GEMM_SPLITK");
        g_splitk_policy = e ? atoi(e) : 1;
    }
    if (rows_per_image <= 0 || g_splitk_policy == 0) return 1;
    const int bn = (N % 160 == 0) ? 160 : (N % 128 == 0 ? 128 : (N <= 64 ? 64 : 128));
    const long long t1 = (long long)mve_cdiv(rows_per_image, BM) * mve_cdiv(N, bn);      // tiles of ONE image
    const int nk = (K + BK - 1) / BK;
    // ceil (round 6): an image of 40 tiles (the 30 x 20 level of Zero123++'s 120 x 80 latent: 80 blocks for a CFG pair, each walking K = 11 520 alone --
    // 271 us per conv, profiles/r06_trace_zero123pp.txt) gets 2 slices instead of 1.  Power-of-two tile counts (every level of a 64 x 64 latent: 64 / 32 /
    // 16 / 8 tiles) divide 64: their slice counts, and with them every bit of those results, are unchanged.
    long long s = (64 + t1 - 1) / t1;
    if (s > nk / 8) s = nk / 8;        // at least 8 K tiles (512 k) per slice
    if (s > 16) s = 16;
    return s < 2 ? 1 : (int)s;
}

int gemm_red_mode();
int* gemm_sk_sync(hipStream_t s);
constexpr int SK_SYNC_TILES = 8192;

// Weight-strip-major block order for launches whose activations are the smaller operand (GemmParams::w_major).  MVE_GEMM_WMAJOR=0 / mve_gemm_deep_tune bit 30 off.
int g_w_major = -1;
bool gemm_w_major_on() {
    if (g_w_major < 0) {
        const char* e = getenv("MVE_GEMM_WMAJOR");
        g_w_major = e ? atoi(e) : 1;
    }
    return g_w_major != 0;
}

// Narrower tile for launches of at most gemm_small_bn_max_blocks() 128 x 160 blocks (see launch_v).  MVE_GEMM_SMALL_BN = 0 | 64 | 128, MVE_GEMM_SMALL_BN_MAX.
int g_small_bn = -1, g_small_bn_max = -1;
int gemm_small_bn() {
    if (g_small_bn < 0) {
        const char* e = getenv("MVE_GEMM_SMALL_BN");
        g_small_bn = e ? atoi(e) : 64;      // same-box sweep, profiles/r06_ab_small_bn.log: Zero123++ 21.55 -> 20.4 ms, 8-image forward -1 %; 128 is neutral
        if (g_small_bn != 64 && g_small_bn != 128) g_small_bn = 0;
    }
    return g_small_bn;
}
int gemm_small_bn_max_blocks() {
    if (g_small_bn_max < 0) {
        const char* e = getenv("MVE_GEMM_SMALL_BN_MAX");
        g_small_bn_max = e ? atoi(e) : 384;
    }
    return g_small_bn_max;
}

int g_small_bm = -1, g_small_bm_max = -1;
int gemm_small_bm() {
    if (g_small_bm < 0) {
        const char* e = getenv("MVE_GEMM_SMALL_BM");
        g_small_bm = e ? atoi(e) : 0;
        if (g_small_bm != 64) g_small_bm = 0;
    }
    return g_small_bm;
}
int gemm_small_bm_max_blocks() {
    if (g_small_bm_max < 0) {
        const char* e = getenv("MVE_GEMM_SMALL_BM_MAX");
        g_small_bm_max = e ? atoi(e) : 512;
    }
    return g_small_bm_max;
}

// K columns per block from which a launch that fills neither 256-row rule takes the ping-pong 256 x 160 tile anyway; 0 = never.  MVE_GEMM_PP160_MINK.
int g_pp160_min_k = -1;
int gemm_pp160_min_k() {
    if (g_pp160_min_k < 0) {
        const char* e = getenv("MVE_GEMM_PP160_MINK");
        g_pp160_min_k = e ? atoi(e) : 1440;      // same-box sweep, profiles/r06_ab_pp160_min_k.log: Zero123++ 22.9 -> 21.4 ms, 8-image forward 12.44 -> 12.18 ms; 640 and below lose again
    }
    return g_pp160_min_k;
}

// The four-stage ring of the 128-row kernel (k_gemm_deep) for launches of at most this many blocks; 0 turns it off.  MVE_GEMM_DEEP / mve_gemm_deep_tune.
int g_gemm_deep = -1;
int gemm_deep_max_blocks() {
    if (g_gemm_deep < 0) {
        const char* e = getenv("MVE_GEMM_DEEP");
        g_gemm_deep = e ? atoi(e) : 0;
    }
    return g_gemm_deep;
}

template <class Tag>
int splitk_reduce_launch(const GemmParams& p, hipStream_t s);

template <class Tag, int BN, int MODE>
int launch_deep(const GemmParams& p, unsigned grid, hipStream_t s) {
    constexpr int SM = gemm_smem_bytes<BN, DEEP_NST>();
    static bool configured[64] = {};
    int dev = 0;
    MVE_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !configured[dev]) {
        MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_deep<Tag, BN, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, SM));
        configured[dev] = true;
    }
    k_gemm_deep<Tag, BN, MODE><<<grid, NT, SM, s>>>(p);
    MVE_LAUNCH_CHECK();
    return splitk_reduce_launch<Tag>(p, s);
}

template <class Tag, int MODE>
int launch_v(const GemmParams& p, hipStream_t s) {
    MVE_CHECK(p.g.kw != 2, MVE_ERR_STATE, "gemm: 2 x 2 conv windows run on the ping-pong kernel only");
    // tile width: prefer the widest tile that divides N (no dead columns), else 128
    int bn = 128;
    if (p.N % 160 == 0) bn = 160;
    else if (p.N % 128 == 0) bn = 128;
    else if (p.N <= 64) bn = 64;
    // (round 6) a launch of at most one 128 x 160 block per CU runs its phases back to back (one wave per SIMD: LDS-DMA, fragment reads + MFMAs and
    // the epilogue add up, profiles/r06_gemm_lab_ablation.txt); narrower tiles put two or more blocks on a CU, whose phases overlap.  gemm_small_bn():
    // the tile width such launches take when it divides N (0 = keep 160).  Bit-identical like every tile choice.
    if (bn == 160 && gemm_small_bn() > 0 && p.N % gemm_small_bn() == 0 &&
        mve_cdiv(p.M, BM) * mve_cdiv(p.N, 160) * (p.splitk > 1 ? p.splitk : 1) <= (unsigned)gemm_small_bn_max_blocks())
        bn = gemm_small_bn();
    const unsigned tiles_m = mve_cdiv(p.M, BM), tiles_n = mve_cdiv(p.N, bn);
    const unsigned grid = tiles_m * tiles_n * (p.splitk > 1 ? p.splitk : 1);
#ifdef MVE_GEMM_LAB
    { const char* e = getenv("MVE_GEMM_LAB_BITS"); const_cast<GemmParams&>(p).dbg = e ? atoi(e) : 0; }
#endif
    // K slices folded inside the launch (gemm_reduce_slices) where every block of the grid is resident at once -- two 72 KiB blocks per CU -- so that a
    // block waiting for its siblings never holds the slot one of them needs: no k_splitk_reduce launch behind such a launch
    GemmParams pf;
    if (p.splitk > 1 && !p.sk_sync && (gemm_red_mode() & 1) && grid <= 512 && tiles_m * tiles_n <= (unsigned)SK_SYNC_TILES &&
        (unsigned long long)p.splitk * p.M * p.N * 4ull < 0xF0000000ull) {
        if (int* sync = gemm_sk_sync(s)) {
            pf = p;
            pf.sk_sync = sync;
            return launch_v<Tag, MODE>(pf, s);
        }
    }
    // launches of at most gemm_deep_max_blocks() blocks (a block or two per CU) and more than two K tiles per block: the four-stage ring (k_gemm_deep)
    const int nk_slice = ((p.K + BK - 1) / BK) / (p.splitk > 1 ? p.splitk : 1);
    if ((int)grid <= gemm_deep_max_blocks() && nk_slice > 2 && bn >= 128) {
        if (bn == 160) return launch_deep<Tag, 160, MODE>(p, grid, s);
        return launch_deep<Tag, 128, MODE>(p, grid, s);
    }
    // ... and 64 x 64 tiles where even the 128 x 64 tiling is at most gemm_small_bm_max_blocks() blocks (MVE_GEMM_SMALL_BM = 64 | 0)
    if (bn == 64 && gemm_small_bm() == 64 && p.N % 64 == 0 && p.M > 64 && (int)grid <= gemm_small_bm_max_blocks()) {
        const unsigned grid64 = mve_cdiv(p.M, 64) * tiles_n * (p.splitk > 1 ? p.splitk : 1);
        k_gemm64<Tag, 64, MODE><<<grid64, NT, 0, s>>>(p);
        MVE_LAUNCH_CHECK();
        return splitk_reduce_launch<Tag>(p, s);
    }
    if (bn == 160) k_gemm<Tag, 160, MODE><<<grid, NT, 0, s>>>(p);
    else if (bn == 128) k_gemm<Tag, 128, MODE><<<grid, NT, 0, s>>>(p);
    else k_gemm<Tag, 64, MODE><<<grid, NT, 0, s>>>(p);
    MVE_LAUNCH_CHECK();
    return splitk_reduce_launch<Tag>(p, s);
}

// What a launch does when the slice rule (choose_splitk) asks for S > 1 slices but the un-split launch already fills the chip (>= one 256 x 320
// tile per CU: 64 images on one GPU at the 32 x 32 and 16 x 16 levels).
//   0 (default, round 4): one block per tile walks all of K in ONE accumulation chain.  The result differs from the sliced sum (what the same
//     image gets in a small batch, where the slices run as separate blocks + reducer) by fp32 summation order only -- well inside the 16-bit output
//     rounding, see tests/test_unet_ops.py::test_unsplit_chain_vs_sliced_sum -- so bitwise batch invariance holds among launches that take the
//     same decision (all small batches; all chip-filling batches), not across the two.
//   1 (MVE_GEMM_STRICT_SPLITK=1 / mve_gemm_tune bit 30): the block emulates the slices (GemmParams::splitk_seq: accumulators folded into an fp32
//     running total at every slice boundary), bitwise equal to split-K + reducer at any batch.  Measured cost at 64 images: the fold drains
//     the DMA ring and moves 160 fp32 registers per lane through HBM per slice: level-1 / level-2 convs 1 080 -> 1 310-1 400 TFLOP/s without it,
//     the N = K GEMMs of level 2 550 -> 900, ff.out 750 -> 1 240; 3.8 ms of a 68 ms step (profiles/r04_oplist_*.log).
int g_strict_splitk = -1;
int gemm_strict_splitk() {
    if (g_strict_splitk < 0) {
        const char* e = getenv("MVE_GEMM_STRICT_SPLITK");
        g_strict_splitk = e ? (atoi(e) != 0) : 0;
    }
    return g_strict_splitk;
}

// minimum number of 256 x 320 blocks for which the big-tile kernel is used (0 disables it); MVE_GEMM_BIG overrides
int g_big_min_blocks = -1;
int g_seq_splitk = 1;         // mve_gemm_tune bit 29 clears it (A/B: real split-K + reducer)
int gemm_big_min_blocks() {
    if (g_big_min_blocks < 0) {
        const char* e = getenv("MVE_GEMM_BIG");
        g_big_min_blocks = e ? atoi(e) : 256;     // one block per CU: measured break-even on MI355X (profiles/r01_ab_gemm_big*.log)
    }
    return g_big_min_blocks;
}

// The 256-row tile has two main loops with bit-identical results: the ping-pong schedule (gemm_pp.hip) wherever it is eligible,
// else the two-stage loop (gemm_big.hip).  mve_gemm_tune bit 27 / MVE_GEMM_PP=0 turn the former off (A/B).
int g_gemm_pp = -1;
bool gemm_pp_on() {
    if (g_gemm_pp < 0) {
        const char* e = getenv("MVE_GEMM_PP");
        g_gemm_pp = e ? atoi(e) : 1;
    }
    return g_gemm_pp != 0;
}
// The two-blocks-per-CU 256 x 160 tile (gemm_pp.hip, NSL = 3) for dense GEMMs, bit-identical results.  2 (default): taken where the dispatcher
// asks for the narrow tile because 320-wide tiles would leave CUs idle (small batches: the per-wave epilogue and the second resident block are
// what those short launches lack); 1: wherever it is eligible (A/B: slower at 64 images); 0: never.  MVE_GEMM_PP2 / mve_gemm_tune.
int g_gemm_pp2 = -1;
int g_old_swizzle = 0;

int gemm_pp2_mode() {
    if (g_gemm_pp2 < 0) {
        const char* e = getenv("MVE_GEMM_PP2");
        g_gemm_pp2 = e ? atoi(e) : 2;
        if (g_gemm_pp2 < 0 || g_gemm_pp2 > 2) g_gemm_pp2 = 2;
    }
    return g_gemm_pp2;
}
// In-kernel slice reduction (round 6; gemm_pp.hip: pp_reduce_slices).  A K-sliced launch that the ping-pong tile can take (320-wide where that fills
// the chip, 160-wide otherwise) folds its slices inside the launch: no k_splitk_reduce launch behind it, and -- for the small launches of a rank
// that holds few images, which used to run on the 128-row two-stage kernel -- the deep LDS-DMA ring of the ping-pong loop.  Bit-identical to
// partials + reducer (same slices, same fold order, same epilogue function).  MVE_GEMM_RED=0 / mve_gemm_red_tune(0) restore the reducer launches.
int g_gemm_red = -1;      // bit 0: the 128-row kernel folds its slices (launch_v);  bit 1: small K-sliced launches go to the ping-pong tile and fold there (launch_red)
int gemm_red_mode() {
    if (g_gemm_red < 0) {
        const char* e = getenv("MVE_GEMM_RED");
        g_gemm_red = e ? atoi(e) : 0;      // off: measured slower than partials + reducer on every K-sliced launch of an 8-image forward but the longest (profiles/r06_fold_modes_8images.log)
    }
    return g_gemm_red;
}
// one zeroed counter array per (device, stream): launches of a stream are ordered, and every launch leaves its counters at zero
int* gemm_sk_sync(hipStream_t s) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int*> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = bufs.find({dev, s});
    if (it != bufs.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    int* b = nullptr;
    if (hipMalloc(&b, SK_SYNC_TILES * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemsetAsync(b, 0, SK_SYNC_TILES * sizeof(int), s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(b); return nullptr; }
    bufs[{dev, s}] = b;
    return b;
}
// -> MVE_OK: launched, result complete;  1: not taken (the caller runs its ordinary path);  < 0: error
int launch_red(int dtype, int mode, const GemmParams& p, hipStream_t s) {
    if (!(gemm_red_mode() & 2) || !gemm_pp_on() || p.splitk <= 1 || p.splitk > 64 || p.M < 64) return 1;
    GemmParams q = p;
    const long long tm = mve_cdiv(p.M, 256);
    if (p.N % 320 == 0 && tm * (p.N / 320) * p.splitk >= 256) q.tile_n = 0;
    else if (p.N % 160 == 0) q.tile_n = 160;
    else return 1;
    const long long tiles = tm * (p.N / (q.tile_n == 160 ? 160 : 320));
    if (tiles > SK_SYNC_TILES || tiles * p.splitk > 256) return 1;      // every block of the grid resident at once (one 104-144 KiB block per CU): a block waiting for its siblings never holds the slot one of them needs
    if ((unsigned long long)p.splitk * p.M * p.N * 4ull >= 0xF0000000ull) return 1;      // the partial tiles are addressed through 32-bit buffer offsets
    q.sk_sync = gemm_sk_sync(s);
    if (!q.sk_sync) return 1;
    return mve_gemm_pp_launch(dtype, mode, &q, s);
}

// MVE_OK after a launch, 1 when neither loop takes the problem (the caller falls back to the 128-row kernel), < 0 on error
int launch_tile256(int dtype, int mode, const GemmParams* q, hipStream_t s) {
    const int pp2 = gemm_pp2_mode();
    // (round 6) in the default mode also the chip-filling launches WITHOUT a GEGLU epilogue: attn1.qkv / attn2.to_q at every level run 8-16 % faster on
    // two 256 x 160 blocks per CU than on one 256 x 320 block (the epilogue of one block under the K loop of the other; K = C is 10-40 steps), the
    // GEGLU launches 6-7 % slower at the 32 x 32 level (profiles/r06_oplist64_pp2_ab.txt) -- residual / pair launches are not eligible for this tile
    if (gemm_pp_on() && mode == 0 && q->splitk <= 1 && q->splitk_seq <= 1 &&
        ((pp2 == 1 && q->tile_n == 0) || (pp2 == 2 && (q->tile_n == 160 || (q->tile_n == 0 && !q->geglu))))) {
        GemmParams r = *q;
        r.tile_n = 161;
        const int rc = mve_gemm_pp_launch(dtype, mode, &r, s);
        if (rc <= 0) return rc;
    }
    if (gemm_pp_on()) {
        const int rc = mve_gemm_pp_launch(dtype, mode, q, s);
        if (rc <= 0) return rc;
    }
    if (q->tile_n == 160 || mve_gemm_big_blocks(q->M, q->N, q->splitk) <= 0) return 1;
    if (q->residual_lo || q->out_lo) return 1;        // residual_pair mode: only the ping-pong 320-wide tile and the 128-row kernel carry the pair
    return mve_gemm_big_launch(dtype, mode, q, s);
}

// Width of the 256-row tile for N columns (0: none fits): 320 (every UNet width), 256 (the VAE's 256 / 512-channel convs; no
// split-K variants), 128 (the VAE's 128-channel convs at image resolution; ping-pong loop only, no split-K).  Any other N -- 8 for
// conv_out -- would leave most of a tile dead and takes the 128 x {64,128,160} kernel, whose results are bit-identical.
int tile256_bn(int N, int splitk) {
    if (N % 320 == 0) return 320;
    if (splitk > 1) return 0;
    if (N % 256 == 0) return 256;
    if (N % 128 == 0 && gemm_pp_on()) return 128;
    return 0;
}
long long tile256_blocks(int M, int N, int splitk) {
    const int bn = tile256_bn(N, splitk);
    if (bn == 0 || M < 64) return 0;
    return (long long)mve_cdiv(M, 256) * (N / bn) * (splitk > 1 ? splitk : 1);
}

template <class Tag, int MODE>
int launch_gemm_split(const GemmParams& p, hipStream_t s);

template <class Tag, int MODE>
int launch_gemm(const GemmParams& p, hipStream_t s) {
    if (tile256_bn(p.N, p.splitk) == 0) return launch_v<Tag, MODE>(p, s);
    // enough 256 x 320 tiles to fill the chip WITHOUT cutting K: one block per tile walks the slices one after the other and
    // reproduces the split-K rounding exactly (GemmParams::splitk_seq) -- no partial tiles, no reducer launch
    if (gemm_big_min_blocks() > 0 && p.splitk > 1 && p.N % 320 == 0 && g_seq_splitk && tile256_blocks(p.M, p.N, 1) >= gemm_big_min_blocks() &&
        (size_t)tile256_blocks(p.M, p.N, 1) * 256 * 320 <= (size_t)p.splitk * p.M * p.N) {
        GemmParams q = p;
        q.splitk_seq = gemm_strict_splitk() ? p.splitk : 0;      // default: ONE accumulation chain over all of K (see gemm_strict_splitk)
        q.splitk = 1;
        const int rc = launch_tile256(Tag::dtype, MODE, &q, s);
        if (rc != 1) return rc;                            // 1: no 256-row loop takes it in this form (e.g. strict slices + residual pair): split K for real below
    }
    // The slice rule asks for more slices than this launch needs to fill the chip (64 images at the 8 x 8 level: 64 tiles x 8 slices): cut K into
    // just enough slices for one block per CU -- every slice fewer is 2 x M x N x 4 bytes of fp32 partials less through HBM and a longer K loop
    // per prologue / epilogue.  Like the single chain above this changes the fp32 summation order with the batch, not the value; the strict mode
    // keeps the rule's slice count.
    GemmParams pfew;
    const GemmParams* pp = &p;
    if (!gemm_strict_splitk() && gemm_big_min_blocks() > 0 && p.splitk > 2 && p.N % 320 == 0) {
        const long long t1 = tile256_blocks(p.M, p.N, 1);
        if (t1 > 0 && t1 * p.splitk >= 2 * gemm_big_min_blocks()) {
            int few = (int)((gemm_big_min_blocks() + t1 - 1) / t1);
            few = few < 2 ? 2 : few;
            if (few < p.splitk) { pfew = p; pfew.splitk = few; pp = &pfew; }
        }
    }
    if (pp != &p) return launch_gemm_split<Tag, MODE>(*pp, s);
    return launch_gemm_split<Tag, MODE>(p, s);
}

template <class Tag, int MODE>
int launch_gemm_split(const GemmParams& p, hipStream_t s) {
    if (p.splitk > 1) {
        const int rc = launch_red(Tag::dtype, MODE, p, s);
        if (rc != 1) return rc;
    }
    // small batches: 256 x 320 tiles would leave CUs without a block, 256 x 160 tiles (ping-pong loop only) still cover them
    const bool narrow = gemm_big_min_blocks() > 0 && gemm_pp_on() && p.splitk <= 1 && p.N % 320 == 0 && p.M >= 64 &&
                        tile256_blocks(p.M, p.N, 1) < gemm_big_min_blocks() && 2 * tile256_blocks(p.M, p.N, 1) >= gemm_big_min_blocks();
    if (narrow || (gemm_big_min_blocks() > 0 && tile256_blocks(p.M, p.N, p.splitk) >= gemm_big_min_blocks())) {
        GemmParams q = p;
        q.tile_n = narrow ? 160 : 0;
        const int rc = launch_tile256(Tag::dtype, MODE, &q, s);
        if (rc < 0) return rc;
        if (rc == 0) {
            if (p.splitk > 1) {
                k_splitk_reduce<Tag><<<mve_cdiv((size_t)p.M * (p.N / 8), 256), 256, 0, s>>>(p);
                MVE_LAUNCH_CHECK();
            }
            return MVE_OK;
        }
    }
    // (round 6) launches too small for either rule above but with a LONG K loop per block: the ping-pong 256 x 160 tile with half the blocks of the
    // 128-row kernel still wins -- its K step costs ~0.45 us per 32 columns against ~1.6 us per 64 for a 128-row block that runs alone on its CU
    // (profiles/r06_gemm_lab_ablation.txt), and the fixed costs of a launch stop mattering.  Zero123++'s CFG pair on a 120 x 80 latent lives here
    // (75-300 blocks of 128 x 160 per conv, K = 2 880 .. 11 520 unsplit or in 2 slices).  Bit-identical like every tile choice.
    if (gemm_pp_on() && gemm_pp160_min_k() > 0 && p.N % 160 == 0 && p.M >= 128 && p.K / (p.splitk > 1 ? p.splitk : 1) >= gemm_pp160_min_k()) {
        GemmParams q = p;
        q.tile_n = 160;
        const int rc = mve_gemm_pp_launch(Tag::dtype, MODE, &q, s);
        if (rc < 0) return rc;
        if (rc == 0) return splitk_reduce_launch<Tag>(p, s);
    }
    return launch_v<Tag, MODE>(p, s);
}

int check_common(const GemmParams& p, const char* who) {
    MVE_CHECK(p.M > 0 && p.N > 0 && p.K > 0, MVE_ERR_ARG, "%s: empty problem M=%d N=%d K=%d", who, p.M, p.N, p.K);
    MVE_CHECK(p.N % 8 == 0 && p.K % 8 == 0, MVE_ERR_ARG, "%s: N (%d) and K (%d) must be multiples of 8", who, p.N, p.K);
    MVE_CHECK(p.W && p.out, MVE_ERR_ARG, "%s: null pointer", who);
    MVE_CHECK(p.ldc % 4 == 0, MVE_ERR_ARG, "%s: ldc must be a multiple of 4", who);
    MVE_CHECK(!p.residual || p.ldr % 8 == 0, MVE_ERR_ARG, "%s: ldr must be a multiple of 8", who);
    MVE_CHECK(!p.rowvec || (p.rows_per_vec > 0 && p.ldrv % 4 == 0 && (p.ldrv >= p.N || p.ldrv == 0)), MVE_ERR_ARG,
              "%s: rowvec needs rows_per_vec > 0 and ldrv (%d) a multiple of 4 >= N (or 0: one vector for all rows)", who, p.ldrv);
    MVE_CHECK(p.ldw % 8 == 0 && p.ldw >= p.K, MVE_ERR_ARG, "%s: ldw (%d) must be a multiple of 8 and >= K", who, p.ldw);
    MVE_CHECK(!(p.geglu && (p.residual || p.out_f32)), MVE_ERR_ARG, "%s: geglu excludes residual/out_f32", who);
    MVE_CHECK(!(p.residual_lo && (!p.residual || p.res_after_scale)), MVE_ERR_ARG, "%s: residual_lo needs a residual that is added before the scale", who);
    MVE_CHECK(!(p.out_lo && (p.out_f32 || p.geglu || p.ldc % 8 != 0 || (p.residual && p.res_after_scale))), MVE_ERR_ARG,
              "%s: out_lo needs a 16-bit, non-GEGLU output with ldc a multiple of 8 and no residual after the scale", who);
    return MVE_OK;
}

// one K-sliced launch of a phase of mve_upsample_conv_phases: launch_gemm's slice policy on the ping-pong kernel alone
template <class Tag>
int launch_phase(GemmParams q, hipStream_t s) {
    const int minb = gemm_big_min_blocks() > 0 ? gemm_big_min_blocks() : 256;
    if (q.splitk > 1 && q.N % 320 != 0) q.splitk = 1;         // (only the 320-wide tile cuts K)
    // (strict mode, MVE_GEMM_STRICT_SPLITK: the rule's slices run as real slices + reducer at any batch -- the path small batches take anyway; the
    // in-block slice emulation of the 3 x 3 convs is not instantiated for this form)
    if (q.splitk > 1 && !gemm_strict_splitk()) {
        const long long t1 = tile256_blocks(q.M, q.N, 1);
        if (t1 >= minb) q.splitk = 1;                         // the un-split launch fills the chip: one accumulation chain
        else if (q.splitk > 2 && t1 > 0 && t1 * q.splitk >= 2 * minb) {
            const int few = (int)((minb + t1 - 1) / t1);
            q.splitk = few < 2 ? 2 : (few < q.splitk ? few : q.splitk);
        }
    }
    if (q.splitk > 1) {
        const int rr = launch_red(Tag::dtype, 1, q, s);
        if (rr != 1) return rr;
    }
    const int rc = mve_gemm_pp_launch(Tag::dtype, 1, &q, s);
    if (rc == 1) {
        mve_set_error("upsample_conv_phases: the ping-pong kernel does not take M=%d N=%d K=%d (mve_upsample_conv_phases_supported)", q.M, q.N, q.K);
        return MVE_ERR_ARG;
    }
    if (rc != MVE_OK) return rc;
    if (q.splitk > 1) {
        k_splitk_reduce<Tag><<<mve_cdiv((size_t)q.M * (q.N / 8), 256), 256, 0, s>>>(q);
        MVE_LAUNCH_CHECK();
    }
    return MVE_OK;
}

// one thread per destination element [phase][o][slab][a][b][c]: the 3 x 3 taps (rows ys, columns xs) that land on window position (a, b) of
// phase (py, px) are summed in fp32 in ascending (row, column) order -- py = 0: rows {0}, {1, 2}; py = 1: rows {0, 1}, {2}; columns likewise
template <class SrcT, class Tag>
__global__ __launch_bounds__(256) void k_pack_phase_weights(const SrcT* __restrict__ w, typename Tag::T* __restrict__ w4, int Cout, int C) {
#pragma clang fp contract(off)
    const size_t per_phase = (size_t)Cout * 4 * C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 4 * per_phase) return;
    const int ph = (int)(i / per_phase);
    size_t r = i - (size_t)ph * per_phase;
    const int o = (int)(r / (4 * C));
    r -= (size_t)o * 4 * C;
    const int slab = (int)(r / 256), t = (int)(r % 256) / 64, c = (int)(r % 64);
    const int py = ph >> 1, px = ph & 1, a = t >> 1, b = t & 1;
    const int y0 = a == 0 ? 0 : (py ? 2 : 1), y1 = a == 0 ? (py ? 1 : 0) : 2;
    const int x0 = b == 0 ? 0 : (px ? 2 : 1), x1 = b == 0 ? (px ? 1 : 0) : 2;
    const SrcT* src = w + ((size_t)o * C + slab * 64 + c) * 9;
    float acc = 0.f;
    for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) acc += (float)src[yy * 3 + xx];
    w4[i] = Tag::from_f32(acc);
}
}  // namespace

extern "C" {

int mve_gemm_tune(int big_min_blocks) {
    // the whole previous word comes back (threshold + option bits), so that old = tune(x); ...; tune(old) restores every switch
    const int old = gemm_big_min_blocks() | (g_seq_splitk ? 0 : (1 << 29)) | (gemm_pp_on() ? 0 : (1 << 27)) | (gemm_pp2_mode() == 1 ? (1 << 26) : 0) | (gemm_pp2_mode() == 0 ? (1 << 28) : 0) |
                    (g_old_swizzle ? (1 << 25) : 0) | (gemm_strict_splitk() ? (1 << 30) : 0);
    if (big_min_blocks >= 0) {
        g_strict_splitk = (big_min_blocks >> 30) & 1;
        g_seq_splitk = (big_min_blocks & (1 << 29)) ? 0 : 1;
        g_gemm_pp = (big_min_blocks & (1 << 27)) ? 0 : 1;
        g_gemm_pp2 = (big_min_blocks & (1 << 26)) ? 1 : ((big_min_blocks & (1 << 28)) ? 0 : 2);
        g_old_swizzle = (big_min_blocks >> 25) & 1;
        mve_gemm_pp_old_swizzle(g_old_swizzle);
        g_big_min_blocks = big_min_blocks & ~((7 << 28) | (1 << 27) | (1 << 26) | (1 << 25));       // (7 << 28): bits 28, 29 and 30
    }
    return old;
}

int mve_gemm_deep_tune(int max_blocks) {
    const int old = gemm_deep_max_blocks() | (gemm_w_major_on() ? 0 : (1 << 30));
    if (max_blocks >= 0) { g_gemm_deep = max_blocks & ~(1 << 30); g_w_major = (max_blocks & (1 << 30)) ? 0 : 1; }
    return old;
}

int mve_gemm_red_tune(int on) {
    const int old = gemm_red_mode();
    if (on >= 0) g_gemm_red = on & 3;
    return old;
}

/* The number of K slices a launch of this shape actually runs with under the current switches (diagnostics / tests; no GPU touched): the slice
 * rule's count (choose_splitk: a function of rows per image, N, K only), 1 where the un-split launch fills the chip (one accumulation chain;
 * the strict mode emulates the rule's slices inside one block instead: reported as the rule's count), or the smallest count that fills the
 * chip where the rule would over-fill it. */
int mve_gemm_effective_splitk(int M, int N, int K, int rows_per_image) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    const int sk = choose_splitk(rows_per_image, N, K);
    if (sk <= 1 || tile256_bn(N, sk) == 0 || gemm_big_min_blocks() <= 0 || N % 320 != 0) return sk < 1 ? 1 : sk;
    const long long t1 = tile256_blocks(M, N, 1);
    if (g_seq_splitk && t1 >= gemm_big_min_blocks() && (size_t)t1 * 256 * 320 <= (size_t)sk * M * N) return gemm_strict_splitk() ? sk : 1;
    if (!gemm_strict_splitk() && sk > 2 && t1 > 0 && t1 * sk >= 2 * gemm_big_min_blocks()) {
        int few = (int)((gemm_big_min_blocks() + t1 - 1) / t1);
        few = few < 2 ? 2 : few;
        return few < sk ? few : sk;
    }
    return sk;
}

size_t mve_gemm_workspace_bytes(int M, int N, int K, int rows_per_image) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int sk = choose_splitk(rows_per_image, N, K);
    return sk > 1 ? (size_t)sk * M * N * sizeof(float) : 0;
}

int mve_gemm(int dtype, const void* A, int lda, const void* W, int ldw, void* out, int ldc, int M, int N, int K,
             const float* bias, const float* rowvec, int ldrv, int rows_per_vec, const void* residual, int ldr, int flags,
             float out_scale, void* workspace, size_t workspace_bytes, int rows_per_image, void* stream) {
    return mve_gemm_pair(dtype, A, lda, W, ldw, out, ldc, M, N, K, bias, rowvec, ldrv, rows_per_vec, residual, ldr, flags, out_scale, workspace,
                         workspace_bytes, rows_per_image, nullptr, nullptr, stream);
}

static int gemm_pair_impl(int dtype, const void* A, int lda, const void* W, int ldw, void* out, int ldc, int M, int N, int K,
                          const float* bias, const float* rowvec, int ldrv, int rows_per_vec, const void* residual, int ldr, int flags,
                          float out_scale, void* workspace, size_t workspace_bytes, int rows_per_image, const void* residual_lo, void* out_lo,
                          void* stream, void* ln_out, int ld_ln, const float* ln_gamma, const float* ln_beta, float ln_eps);

int mve_gemm_pair(int dtype, const void* A, int lda, const void* W, int ldw, void* out, int ldc, int M, int N, int K,
                  const float* bias, const float* rowvec, int ldrv, int rows_per_vec, const void* residual, int ldr, int flags,
                  float out_scale, void* workspace, size_t workspace_bytes, int rows_per_image, const void* residual_lo, void* out_lo,
                  void* stream) {
    return gemm_pair_impl(dtype, A, lda, W, ldw, out, ldc, M, N, K, bias, rowvec, ldrv, rows_per_vec, residual, ldr, flags, out_scale, workspace,
                          workspace_bytes, rows_per_image, residual_lo, out_lo, stream, nullptr, 0, nullptr, nullptr, 0.f);
}

/* mve_gemm_pair followed by LayerNorm of the output rows: d_ln_out[m] = LayerNorm(out[m]) * gamma + beta over the N columns, where out[m] is the row
 * as a consumer reads it back (hi + lo8 when d_out_lo is given, the 16-bit row otherwise).  Where the launch runs on the 320-wide pair tile
 * (N = 320, whole 256-row tiles, bias, no K slices: the residual-stream GEMMs of the 64 x 64 level from 16 images up) the tile that produces a row
 * normalises it in its epilogue -- the row never comes back from HBM for its LayerNorm; every other launch is followed by the LayerNorm kernel.
 * Same row arithmetic either way (csrc/ln_core.h): bit-identical.  (BasicTransformerBlock: norm1 / norm2 / norm3 behind proj_in / attn1.to_out /
 * attn2.to_out, diffusers 0.27.2 as driven from lib/models/architecture/diffusers.py:69-97.) */
int mve_gemm_pair_ln(int dtype, const void* A, int lda, const void* W, int ldw, void* out, int ldc, int M, int N, int K,
                     const float* bias, const void* residual, int ldr, void* workspace, size_t workspace_bytes, int rows_per_image,
                     const void* residual_lo, void* out_lo, void* ln_out, int ld_ln, const float* ln_gamma, const float* ln_beta, float ln_eps,
                     void* stream) {
    MVE_CHECK(ln_out && ln_gamma && ln_beta && ld_ln >= N && ld_ln % 8 == 0, MVE_ERR_ARG, "gemm_pair_ln: LayerNorm output / parameters missing or ld_ln (%d) bad", ld_ln);
    (void)mve_gemm_pp_ln_fused();                 // clear a stale flag
    const int rc = gemm_pair_impl(dtype, A, lda, W, ldw, out, ldc, M, N, K, bias, nullptr, 0, 0, residual, ldr, 0, 1.0f, workspace, workspace_bytes,
                                  rows_per_image, residual_lo, out_lo, stream, ln_out, ld_ln, ln_gamma, ln_beta, ln_eps);
    if (rc != MVE_OK || M == 0) return rc;
    if (mve_gemm_pp_ln_fused()) return MVE_OK;
    return mve_layernorm_pair(dtype, out, ldc, ln_out, ld_ln, M, N, ln_gamma, ln_beta, ln_eps, out_lo, stream);
}

int mve_gemm_ln_fuse_tune(int on) { return mve_gemm_pp_ln_fuse_tune(on); }

static int gemm_pair_impl(int dtype, const void* A, int lda, const void* W, int ldw, void* out, int ldc, int M, int N, int K,
                          const float* bias, const float* rowvec, int ldrv, int rows_per_vec, const void* residual, int ldr, int flags,
                          float out_scale, void* workspace, size_t workspace_bytes, int rows_per_image, const void* residual_lo, void* out_lo,
                          void* stream, void* ln_out, int ld_ln, const float* ln_gamma, const float* ln_beta, float ln_eps) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.ln_out = ln_out; p.ld_ln = ld_ln; p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_eps = ln_eps;
    p.A = A; p.W = W; p.out = out; p.bias = bias; p.rowvec = rowvec; p.residual = residual;
    p.residual_lo = residual_lo; p.out_lo = out_lo;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldrv = ldrv;
    p.rows_per_vec = rows_per_vec;
    p.geglu = (flags & MVE_GEMM_GEGLU) ? 1 : 0;
    p.out_f32 = (flags & MVE_GEMM_OUT_F32) ? 1 : 0;
    p.out_scale = out_scale;
    p.res_after_scale = (flags & MVE_GEMM_RES_AFTER_SCALE) ? 1 : 0;
    if (M == 0) return MVE_OK;
    int rc = check_common(p, "gemm");
    if (rc) return rc;
    MVE_CHECK(A && lda % 8 == 0 && lda >= K, MVE_ERR_ARG, "gemm: bad A/lda (%d)", lda);
    p.w_major = gemm_w_major_on() && M < N;       // fewer activation rows than weight rows: walk the row panels inside a weight strip (GemmParams::w_major)
    p.splitk = 1;
    if (workspace && !(flags & MVE_GEMM_NO_SPLITK)) {
        const int sk = choose_splitk(rows_per_image, N, K);
        if (sk > 1 && workspace_bytes >= (size_t)sk * M * N * sizeof(float)) { p.splitk = sk; p.partial = (float*)workspace; }
    }
    if (dtype == MVE_F16) return launch_gemm<F16Tag, 0>(p, (hipStream_t)stream);
    if (dtype == MVE_BF16) return launch_gemm<BF16Tag, 0>(p, (hipStream_t)stream);
    mve_set_error("gemm: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

static int conv3x3_impl(int dtype, const void* x1, int C1, const void* x2, int C2, const void* x3, int C3, const void* x4, int C4, int B,
                        int Hs, int Ws, int stride, int upsample, const void* W, int Cout, void* out, int ldc, const float* bias,
                        const float* rowvec, int ldrv, const void* residual, int ldr, int flags, float out_scale, void* workspace,
                        size_t workspace_bytes, void* stream, const void* residual_lo = nullptr, void* out_lo = nullptr) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    MVE_CHECK(stride == 1 || stride == 2, MVE_ERR_ARG, "conv3x3: stride must be 1 or 2");
    MVE_CHECK(!(upsample && stride != 1), MVE_ERR_ARG, "conv3x3: upsample requires stride 1");
    MVE_CHECK(C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0, MVE_ERR_ARG,
              "conv3x3: channel counts must be multiples of 8 (C1=%d C2=%d)", C1, C2);
    MVE_CHECK(x1 && (C2 == 0 || x2), MVE_ERR_ARG, "conv3x3: null input");
    if (B == 0) return MVE_OK;
    p.g.Hs = Hs; p.g.Ws = Ws;
    p.g.ups = upsample ? 1 : 0;
    p.g.Hv = Hs << p.g.ups; p.g.Wv = Ws << p.g.ups;
    p.g.stride = stride;
    p.g.pad = p.g.pad_x = (flags & MVE_CONV_PAD_BR) ? 0 : 1;        // one output size either way: the window only shifts by a pixel
    MVE_CHECK(p.g.pad == 1 || (stride == 2 && !upsample && Hs % 2 == 0 && Ws % 2 == 0), MVE_ERR_ARG,
              "conv3x3: MVE_CONV_PAD_BR is the stride-2 downsampler of an even-sized input");
    p.g.Ho = (p.g.Hv + 2 - 3) / stride + 1;
    p.g.Wo = (p.g.Wv + 2 - 3) / stride + 1;
    p.g.C1 = C1; p.g.C2 = C2;
    p.g.chunk64 = (flags & MVE_CONV_W_CHUNK64) ? 1 : 0;
    MVE_CHECK(!p.g.chunk64 || (C1 % 64 == 0 && C2 % 64 == 0), MVE_ERR_ARG,
              "conv3x3: MVE_CONV_W_CHUNK64 needs channel counts that are multiples of 64 (C1=%d C2=%d)", C1, C2);
    p.A = x1; p.A2 = x2; p.W = W; p.out = out; p.bias = bias; p.rowvec = rowvec; p.residual = residual;
    p.residual_lo = residual_lo; p.out_lo = out_lo;
    p.M = B * p.g.Ho * p.g.Wo; p.N = Cout; p.K = 9 * (C1 + C2);
    if (C3 > 0) {     // fused 1x1 shortcut: K continues over the channels of x3 (and x4)
        MVE_CHECK(p.g.chunk64 && !upsample && stride == 1 && C3 % 64 == 0 && C4 >= 0 && C4 % 64 == 0 && x3 && (C4 == 0 || x4), MVE_ERR_ARG,
                  "conv3x3_shortcut: needs MVE_CONV_W_CHUNK64, stride 1, no upsample and shortcut channel counts that are multiples of 64");
        p.g.nk_main = p.K / BK;
        p.g.C3 = C3; p.g.C4 = C4;
        p.A3 = x3; p.A4 = x4;
        p.K += C3 + C4;
    }
    p.ldc = ldc; p.ldr = ldr; p.ldrv = ldrv; p.ldw = p.K;
    p.rows_per_vec = p.g.Ho * p.g.Wo;
    p.geglu = 0;
    p.out_f32 = (flags & MVE_GEMM_OUT_F32) ? 1 : 0;
    p.out_scale = out_scale;
    p.res_after_scale = (flags & MVE_GEMM_RES_AFTER_SCALE) ? 1 : 0;
    int rc = check_common(p, "conv3x3");
    if (rc) return rc;
    p.w_major = gemm_w_major_on() && (long long)p.M * (C1 + C2) < (long long)p.N * p.K;      // activation tensor smaller than the weights (GemmParams::w_major)
    p.splitk = 1;
    if (workspace && !(flags & MVE_GEMM_NO_SPLITK)) {
        const int sk = choose_splitk(p.g.Ho * p.g.Wo, p.N, p.K);
        if (sk > 1 && workspace_bytes >= (size_t)sk * p.M * p.N * sizeof(float)) { p.splitk = sk; p.partial = (float*)workspace; }
    }
    if (dtype == MVE_F16) return launch_gemm<F16Tag, 1>(p, (hipStream_t)stream);
    if (dtype == MVE_BF16) return launch_gemm<BF16Tag, 1>(p, (hipStream_t)stream);
    mve_set_error("conv3x3: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

int mve_conv3x3(int dtype, const void* x1, int C1, const void* x2, int C2, int B, int Hs, int Ws, int stride,
                int upsample, const void* W, int Cout, void* out, int ldc, const float* bias, const float* rowvec,
                int ldrv, const void* residual, int ldr, int flags, float out_scale, void* workspace, size_t workspace_bytes,
                void* stream) {
    return conv3x3_impl(dtype, x1, C1, x2, C2, nullptr, 0, nullptr, 0, B, Hs, Ws, stride, upsample, W, Cout, out, ldc, bias, rowvec, ldrv,
                        residual, ldr, flags, out_scale, workspace, workspace_bytes, stream);
}

int mve_conv3x3_pair(int dtype, const void* x1, int C1, const void* x2, int C2, int B, int Hs, int Ws, int stride,
                     int upsample, const void* W, int Cout, void* out, int ldc, const float* bias, const float* rowvec,
                     int ldrv, const void* residual, int ldr, int flags, float out_scale, void* workspace, size_t workspace_bytes,
                     const void* residual_lo, void* out_lo, void* stream) {
    return conv3x3_impl(dtype, x1, C1, x2, C2, nullptr, 0, nullptr, 0, B, Hs, Ws, stride, upsample, W, Cout, out, ldc, bias, rowvec, ldrv,
                        residual, ldr, flags, out_scale, workspace, workspace_bytes, stream, residual_lo, out_lo);
}

/* Nearest-2x upsample + 3 x 3 conv (pad 1) as FOUR 2 x 2 convs over the low-resolution input, one per output parity (py, px): output pixel
 * (2 i + py, 2 j + px) sees source rows {i - 1 + py, i + py} and columns {j - 1 + px, j + px} only, because the 3 x 3 taps that fall on the same
 * source pixel add up: 4 / 9 of the multiply-adds of the conv over the upsampled image, and the zero padding of the upsampled image is the zero
 * padding of the source.  W4 = the summed taps [phase = 2 py + px][Cout][C / 64][2][2][64] (mve_pack_upsample_phase_weights: summed in fp32, ONE
 * rounding to the storage type -- a rounding the 3 x 3 form does not have: the two forms agree to the storage precision of the weights, not bit
 * for bit).  A phase is a launch of the ping-pong kernel with a 2 x 2 window (ConvGeom::kw) that writes its quarter of the [B][2H][2W][Cout]
 * output through the grouped output rows of GemmParams::orow_*; where a phase fills whole 256-row tiles the four phases are ONE launch
 * (ConvGeom::phase_rows).  K slices: launch_gemm's policy with the image's 4 Hs Ws rows as the rule's rows per image. */

int mve_pack_upsample_phase_weights(int src_dtype, int dst_dtype, const void* w_oihw, int Cout, int C, void* W4, void* stream) {
    MVE_CHECK(w_oihw && W4 && Cout > 0 && C > 0 && C % 64 == 0, MVE_ERR_ARG, "pack_upsample_phase_weights: needs C %% 64 == 0 (Cout=%d C=%d)", Cout, C);
    const size_t n = (size_t)16 * Cout * C;
    const unsigned grid = (unsigned)mve_cdiv(n, 256);
    hipStream_t s = (hipStream_t)stream;
#define MVE_PPW(ST, TAG) k_pack_phase_weights<ST, TAG><<<grid, 256, 0, s>>>((const ST*)w_oihw, (typename TAG::T*)W4, Cout, C)
    if (dst_dtype == MVE_F16) {
        if (src_dtype == MVE_F32) MVE_PPW(float, F16Tag); else if (src_dtype == MVE_F16) MVE_PPW(f16, F16Tag); else if (src_dtype == MVE_BF16) MVE_PPW(bf16, F16Tag);
        else { mve_set_error("pack_upsample_phase_weights: unsupported source dtype %d", src_dtype); return MVE_ERR_ARG; }
    } else if (dst_dtype == MVE_BF16) {
        if (src_dtype == MVE_F32) MVE_PPW(float, BF16Tag); else if (src_dtype == MVE_F16) MVE_PPW(f16, BF16Tag); else if (src_dtype == MVE_BF16) MVE_PPW(bf16, BF16Tag);
        else { mve_set_error("pack_upsample_phase_weights: unsupported source dtype %d", src_dtype); return MVE_ERR_ARG; }
    } else { mve_set_error("pack_upsample_phase_weights: unsupported destination dtype %d", dst_dtype); return MVE_ERR_ARG; }
#undef MVE_PPW
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

int mve_upsample_conv_phases_supported(int C, int Cout, int B, int Hs, int Ws) {
    if (C <= 0 || C % 64 != 0 || Cout <= 0 || B <= 0 || Hs <= 0 || Ws <= 0 || (Ws & (Ws - 1)) != 0) return 0;
    if (Cout % 320 != 0 && Cout % 256 != 0 && Cout % 128 != 0) return 0;
    if (!gemm_pp_on() || (long long)B * Hs * Ws < 64 || (long long)B * Hs * Ws > 0x7fffffffll / 4) return 0;
    // the ping-pong kernel addresses its source and its weights through 32-bit buffer offsets (gemm_pp.hip: pp_fits / pp_eligible): a larger
    // launch (a VAE decode of >= 128 images at the 256-channel upsampler) keeps the 3 x 3 form, which falls back to the other 256-row loops
    const unsigned long long lim = 0xFFFFFF00ull - 65536ull;
    if (((unsigned long long)B * Hs * Ws + 2ull * Ws + 4) * (unsigned long long)C * 2 >= lim) return 0;
    if ((unsigned long long)Cout * 4 * C * 2 >= lim) return 0;
    return 1;
}

/* 1 (default; MVE_PHASES_ONE_LAUNCH): the four phases run as one launch where a phase fills whole tiles; 0: always four launches.  Negative: query.
 * Returns the previous value. */
int mve_upsample_conv_phases_tune(int one_launch) {
    static int cur = -1;
    if (cur < 0) { const char* e = getenv("MVE_PHASES_ONE_LAUNCH"); cur = e ? (atoi(e) != 0) : 1; }
    const int old = cur;
    if (one_launch >= 0) cur = one_launch ? 1 : 0;
    return old;
}

size_t mve_upsample_conv_phases_workspace_bytes(int C, int Cout, int B, int Hs, int Ws) {
    return mve_gemm_workspace_bytes(4 * B * Hs * Ws, Cout, 4 * C, 4 * Hs * Ws);      // (the four phases of an image are 4 Hs Ws rows of one launch)
}

int mve_upsample_conv_phases(int dtype, const void* x, int C, int B, int Hs, int Ws, const void* W4, int Cout, void* out, const float* bias,
                             int flags, void* workspace, size_t workspace_bytes, void* out_lo, void* stream) {
    if (B == 0) return MVE_OK;
    MVE_CHECK(mve_upsample_conv_phases_supported(C, Cout, B, Hs, Ws), MVE_ERR_ARG,
              "upsample_conv_phases: needs C %% 64 == 0, Cout a multiple of 128, a power-of-two source width and >= 64 source pixels (C=%d Cout=%d B=%d %dx%d)",
              C, Cout, B, Hs, Ws);
    MVE_CHECK(x && W4 && out, MVE_ERR_ARG, "upsample_conv_phases: null pointer");
    MVE_CHECK(dtype == MVE_F16 || dtype == MVE_BF16, MVE_ERR_ARG, "upsample_conv_phases: unsupported dtype %d", dtype);
    int lw = 0;
    while ((1 << lw) < Ws) ++lw;
    // All four phases in ONE launch when the rows of a phase fill whole 256-row tiles (ConvGeom::phase_rows): the phases then share the chip
    // like the tiles of any conv (64 images at the 8 x 8 level: 4 x 64 tiles = one block per CU in a single accumulation chain, instead of four
    // launches of 64 tiles x 4 K slices and four reducers; 8 images per rank: 2 launches instead of 8).  Same arithmetic per output element either
    // way up to the K-slice policy.  MVE_PHASES_ONE_LAUNCH=0: always four launches (A/B).
    const bool fused = mve_upsample_conv_phases_tune(-1) && (B * Hs * Ws) % 256 == 0;
    for (int ph = 0; ph < (fused ? 1 : 4); ++ph) {
        const int py = ph >> 1, px = ph & 1;
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.g.Hs = p.g.Hv = p.g.Ho = Hs;
        p.g.Ws = p.g.Wv = p.g.Wo = Ws;
        p.g.stride = 1;
        p.g.kw = 2;
        p.g.pad = 1 - py; p.g.pad_x = 1 - px;
        p.g.C1 = C;
        p.g.chunk64 = 1;
        p.M = B * Hs * Ws; p.N = Cout; p.K = 4 * C;
        if (fused) { p.g.phase_rows = p.M; p.M *= 4; }
        p.A = x;
        p.W = (const char*)W4 + (size_t)ph * Cout * p.K * 2;
        const size_t o0 = ((size_t)py * 2 * Ws + px) * Cout * 2;       // bytes: pixel (py, px) of image 0
        p.out = (char*)out + o0;
        p.out_lo = out_lo ? (char*)out_lo + o0 / 2 : nullptr;         // (lo8: one byte per element)
        p.bias = bias;
        p.ldc = 2 * Cout; p.ldw = p.K;
        p.orow_shift = lw; p.orow_extra = 2 * Ws * Cout;
        p.rows_per_vec = Hs * Ws;
        p.out_scale = 1.0f;
        int rc = check_common(p, "upsample_conv_phases");
        if (rc) return rc;
        p.splitk = 1;
        if (workspace && !(flags & MVE_GEMM_NO_SPLITK)) {
            // The slice rule sees an image as the 4 Hs Ws rows its four phases put into a launch (whether or not they share one): K = 4 C is short
            // and the launch is four times an ordinary conv's rows, so the rule stops slicing from the 16 x 16 level up -- and a view gets the same
            // slices alone or in a batch until the batch fills the chip (tests/test_unet.py::test_engine_sd15_full_size_properties).
            const int sk = choose_splitk(4 * Hs * Ws, p.N, p.K);
            if (sk > 1 && workspace_bytes >= (size_t)sk * p.M * p.N * sizeof(float)) { p.splitk = sk; p.partial = (float*)workspace; }
        }
        rc = dtype == MVE_F16 ? launch_phase<F16Tag>(p, (hipStream_t)stream) : launch_phase<BF16Tag>(p, (hipStream_t)stream);
        if (rc) return rc;
    }
    return MVE_OK;
}

int mve_conv3x3_shortcut_pair(int dtype, const void* x1, int C1, const void* x3, int C3, const void* x4, int C4, int B, int Hs, int Ws,
                              const void* W, int Cout, void* out, int ldc, const float* bias, const float* bias2, int flags, float out_scale,
                              void* workspace, size_t workspace_bytes, void* out_lo, void* stream) {
    MVE_CHECK(C3 > 0, MVE_ERR_ARG, "conv3x3_shortcut: no shortcut source");
    return conv3x3_impl(dtype, x1, C1, nullptr, 0, x3, C3, x4, C4, B, Hs, Ws, 1, 0, W, Cout, out, ldc, bias, bias2, 0, nullptr, 0,
                        flags | MVE_CONV_W_CHUNK64, out_scale, workspace, workspace_bytes, stream, nullptr, out_lo);
}

int mve_conv3x3_shortcut(int dtype, const void* x1, int C1, const void* x3, int C3, const void* x4, int C4, int B, int Hs, int Ws,
                         const void* W, int Cout, void* out, int ldc, const float* bias, const float* bias2, const void* residual, int ldr,
                         int flags, float out_scale, void* workspace, size_t workspace_bytes, void* stream) {
    MVE_CHECK(C3 > 0, MVE_ERR_ARG, "conv3x3_shortcut: no shortcut source");
    // bias2 (the shortcut's bias) rides in the per-image row-vector slot with a zero stride: every row adds the same vector
    return conv3x3_impl(dtype, x1, C1, nullptr, 0, x3, C3, x4, C4, B, Hs, Ws, 1, 0, W, Cout, out, ldc, bias, bias2, 0, residual, ldr,
                        flags | MVE_CONV_W_CHUNK64, out_scale, workspace, workspace_bytes, stream);
}

}  // extern "C"
