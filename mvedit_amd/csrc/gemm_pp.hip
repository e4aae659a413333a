// Ping-pong main loop for the 256-row GEMM / implicit-GEMM conv tile (same tile, wave arrangement, MFMA order and epilogue as
// gemm_big.hip, hence bit-identical results; only the schedule of one K step differs).  Tile widths 320 / 256 (waves 2 x 4) and
// 160 / 128 (waves 4 x 2); the description below is for 320.
//
// Why: rocprofv3 PMC on k_gemm_big (profiles/r02_rocprof_v1_summary.txt) shows the MFMA pipe 47 % busy at the clock the kernel
// actually runs at.  Its 8 waves run one K tile in lock-step -- all of them issue their 9 LDS-DMA pieces, then all of them read
// fragments, then all of them issue MFMAs -- so the two waves that share a SIMD want the matrix pipe at the same time and leave it
// idle at the same time.  Here the two wave groups (waves 0-3 and 4-7, one wave of each per SIMD) run half a step apart, held there
// by a barrier after every section:
//
//     interval   2k        2k+1      2k+2      2k+3
//     group 0    L(k)      M(k)      L(k+1)    M(k+1)          L(k): 13 ds_read_b128 of step k's fragments, address part of step k+3
//     group 1    M(k-1)    L(k)      M(k)      L(k+1)          M(k): 40 MFMA 16x16x32, this wave's LDS-DMA pieces of step k+3 between them
//
// so one wave of every SIMD is always in its MFMA section while the other one does the LDS work.
//   * K step = 32 (one MFMA k-step): only 13 fragments (52 VGPRs) are live next to the 160 accumulator registers.
//   * LDS: a ring of 4 slots of 36 KiB (256 + 320 rows x 64 B) = 144 KiB, the same footprint as the two 72 KiB stages.  A slot
//     row holds the 4 16-byte chunks of one (row, K step) XOR-swizzled by (row >> 1) & 3: a ds_read_b128 lane group -- the hardware's groups
//     are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md), not 16 consecutive lanes -- touches all
//     16 slots of a bank row exactly once.  The swizzle is applied on the global side of the LDS-DMA (lane -> source chunk).
//   * prefetch distance 3 steps, never drained: the pieces of step k+3 leave inside M(k) (one behind each group of 8 MFMAs, M0
//     written one MFMA earlier: ~10 cycles per piece there against 60-180 in an L section busy with LDS reads); at the end of L(k) a
//     wave waits (counted vmcnt) for ITS pieces of step k+1 only -- issued in M(k-2) -- and the barrier that ends the interval
//     publishes them.  Hazards, by interval number:
//       RAW  step k+1 is read in 2k+2 (group 0) and 2k+3 (group 1); its B pieces were waited for by group 0 before the barrier
//            ending 2k, its A pieces by group 1 before the barrier ending 2k+1.
//       WAR  step k+3 goes to slot (k-1) & 3, last read in 2k-2 / 2k-1 with lgkmcnt(0) before the barrier ending 2k-1; the
//            earliest issue is group 0's in 2k+1.
//   * role split: group 0 streams the weight rows (5 pieces per wave and step for BN = 320), group 1 the activation rows (4 pieces):
//     a wave carries the address state of one operand only, and each role has its own straight-line copy of the loop.
//   * addresses: buffer_load_dwordx4 ... lds with the step's uniform part (K offset, conv tap and channel slab) folded into the
//     resource base by SALU and a per-lane byte offset that is constant (GEMM, weights) or changes with the conv's source tensor
//     only.  Halo pixels use an all-ones offset, steps past the end of K a zero-sized resource: the DMA writes zeros.
//   * LDS-DMA, waits and MFMAs are inline asm: hipcc would drain vmcnt to 0 at every barrier and before LDS reads it cannot
//     disambiguate, and it rotates the accumulators through the loop when the MFMAs are builtins.  tools/check_pp_isa.py (a CPU
//     test) verifies on the device assembly that the compiler adds no spill, wait, vector memory instruction or M0 user to the loops.
//   * section timing: tools/pp_profile.py (mve_gemm_pp_profile).
#include "common.h"

#include <type_traits>

#include "gemm_shared.h"
#include "gemm_big_epilogue.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int PBM = 256, PNTH = 512;
constexpr int PROWB = 64;                               // bytes per row and K step
constexpr int PNSLOT = 4;
constexpr int P_A_SLOT = PBM * PROWB;                   // 16 KiB
constexpr unsigned P_NUMREC = 0xFFFFFF00u;              // resource size: every valid offset is below it, the all-ones halo offset above
constexpr int pp_slot_bytes(int bn) { return P_A_SLOT + bn * PROWB; }
constexpr int pp_smem(int bn, int nsl = PNSLOT) { return nsl * pp_slot_bytes(bn) + (bn == 160 ? 1024 : 0); }      // 160: + a 1 KiB dump for the filler piece
static_assert(2 * pp_smem(160, 3) <= 160 * 1024, "two blocks of the three-slot 256 x 160 tile per CU");
static_assert(64 * (320 + 4) * 4 <= pp_smem(320) && 64 * (256 + 4) * 4 <= pp_smem(256) && 64 * (128 + 4) * 4 <= pp_smem(128) &&
              64 * (160 + 4) * 4 <= pp_smem(160), "epilogue staging must fit");

__device__ __forceinline__ i32x4 pp_rsrc(unsigned long long base, bool live) {
    i32x4 r;          // readfirstlane: the operands are wave-uniform by construction; this pins them to SGPRs for the asm below
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane(live ? (int)P_NUMREC : 0);
    r[3] = 0x00020000;
    return r;
}

// One LDS-DMA piece of 1 KiB: 64 lanes x 16 B, lane-linear in LDS from the offset held in M0.  The kernel owns M0 from pp_m0_take() to
// pp_m0_give(): nothing hipcc emits for it touches M0 (no movrel, GWS, sendmsg or LDS-DMA builtin), which the build's ISA check
// (tools/check_pp_isa.py) confirms, so the destination is written once per piece, one MFMA ahead of the load (the 1-wait-state
// s_mov m0 -> LDS-DMA hazard is covered by that MFMA; pp_piece_now carries its own s_nop for the prologue).
__device__ __forceinline__ unsigned pp_m0_take() {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0 ; pp_kloop_begin (marker for tools/check_pp_isa.py)" : "=s"(keep));
    return keep;
}
__device__ __forceinline__ void pp_m0_give(unsigned keep) { asm volatile("s_mov_b32 m0, %0 ; pp_kloop_end" ::"s"(keep)); }
__device__ __forceinline__ void pp_set_m0(unsigned dst) { asm volatile("s_mov_b32 m0, %0" ::"s"(dst)); }
__device__ __forceinline__ void pp_dma_m0(unsigned vo, i32x4 rsrc) {
    asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void pp_piece_now(unsigned vo, i32x4 rsrc, unsigned dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo), "s"(rsrc), "s"(dst) : "memory");
}

template <int N>
__device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_barrier_lds() {      // end of an L section: fragments in registers, slot reads retired
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// In-place MFMA (acc += a x b) issued from asm: with the builtin, hipcc rotates the 160 accumulator registers through the unrolled
// ring (destination != C operand on 168 of 240 MFMAs) and spills inside the loop.  The sections are hand-ordered anyway; what the
// compiler no longer sees is the MFMA -> VALU read-after-write distance on the accumulators: pp_mfma_settle() pads it before the
// accumulators are read by ordinary code.
template <class Tag> struct PMfma;
template <> struct PMfma<F16Tag> {
    static __device__ __forceinline__ void run(f32x4& c, const f16x8& a, const f16x8& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
};
template <> struct PMfma<BF16Tag> {
    static __device__ __forceinline__ void run(f32x4& c, const bf16x8& a, const bf16x8& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
};
__device__ __forceinline__ void pp_mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }


// Epilogue of the two-blocks-per-CU tile (NSL = 3: 256 x 160, waves 4 x 2, wave tile 64 x 80).  Every wave finishes its own sub-tile through a
// wave-private LDS window -- no workgroup barrier: a wave's LDS operations execute in order, so its staging writes, the reads behind them and
// the next group's writes need no fence, and the eight waves drift apart instead of meeting four times per tile.  Only what the launcher
// admits (pp2_eligible): 16-bit output, unit output scale, residual (if any) added before it, no per-image row vector, no split K, whole
// tiles.  Same operations in the same order as gemm_epilogue_tail / the GEGLU path of big_tile_epilogue: identical bits.
constexpr int P2_CS_LD = 84;                              // floats per staged row (80 + 4: the 8-lane groups of a ds_write_b128 hit 8 distinct bank quads)
constexpr int P2_CS_WAVE = 16 * P2_CS_LD * 4;             // 16 rows at a time: 5376 B per wave
constexpr int P2_HS_LD = 56;                              // GEGLU: 16-bit elements per staged row (40 + 16: 112-byte rows keep the 16-byte reads aligned)
constexpr int P2_HS_WAVE = 64 * P2_HS_LD * 2;             // all 64 rows of the wave: 7168 B
static_assert(8 * P2_CS_WAVE <= pp_smem(160, 3) && 8 * P2_HS_WAVE <= pp_smem(160, 3), "epilogue staging must fit");

template <class Tag>
__device__ __forceinline__ void pp2_epilogue(const GemmParams& p, f32x4 (&acc)[5][4], unsigned char* smem, int m0, int n0, int wid, int lane, int wm, int wn) {
#pragma clang fp contract(off)
    typedef typename Tag::T T;
    typedef typename Tag::V8 V8;
    constexpr int NF = 5, MF = 4;
    const int mw = m0 + wm * 64, nw = n0 + wn * 80;
    if (p.geglu) {
        // out[m][i] = (v[2i] + b[2i]) * gelu(v[2i+1] + b[2i+1]): a lane owns two (value, gate) pairs of a row per fragment
        typedef T T2 __attribute__((ext_vector_type(2)));
        T* Hs = reinterpret_cast<T*>(smem + wid * P2_HS_WAVE);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int c = j * 16 + (lane >> 4) * 4;
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + nw + c);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const f32x4 v = acc[j][i];
                T2 o;
                o[0] = Tag::from_f32((v[0] + b4[0]) * gelu_erf(v[1] + b4[1]));
                o[1] = Tag::from_f32((v[2] + b4[2]) * gelu_erf(v[3] + b4[3]));
                *reinterpret_cast<T2*>(Hs + (i * 16 + (lane & 15)) * P2_HS_LD + (c >> 1)) = o;
            }
        }
        // 64 rows x 40 outputs = 320 chunks of 8: five per lane, row-contiguous 80-byte segments
        T* outw = reinterpret_cast<T*>(p.out) + (size_t)mw * p.ldc + (nw >> 1);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int task = k * 64 + lane;
            const int r = task / 5, ch = task - r * 5;
            *reinterpret_cast<V8*>(outw + (size_t)r * p.ldc + ch * 8) = *reinterpret_cast<const V8*>(Hs + r * P2_HS_LD + ch * 8);
        }
        return;
    }
    // 16 rows at a time through fp32 staging; lane = (8-column chunk ch of the 10, row r0 of 6), rows r0, r0 + 6, r0 + 12 of a group
    float* Cs = reinterpret_cast<float*>(smem + wid * P2_CS_WAVE);
    const bool act = lane < 60;
    const int ch = act ? lane % 10 : 0, r0 = act ? lane / 10 : 0;
    const int n = nw + ch * 8;
    f32x4 fb0 = f32x4{0.f, 0.f, 0.f, 0.f}, fb1 = fb0;
    if (p.bias) { fb0 = *reinterpret_cast<const f32x4*>(p.bias + n); fb1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
    T* outb = reinterpret_cast<T*>(p.out) + (size_t)(mw + r0) * p.ldc + n;
    const T* resb = reinterpret_cast<const T*>(p.residual) + (size_t)(mw + r0) * p.ldr + n;
    const bool has_res = p.residual != nullptr;
    // row of chunk k relative to r0: the third one exists for r0 < 4 only; the other lanes redo their second row and store nothing
    const int rel2 = r0 < 4 ? 12 : 6;
    auto run = [&](auto res_c) {
        constexpr bool RES = decltype(res_c)::value;
        V8 fr[3];
        if constexpr (RES) {
            fr[0] = *reinterpret_cast<const V8*>(resb);
            fr[1] = *reinterpret_cast<const V8*>(resb + (size_t)6 * p.ldr);
            fr[2] = *reinterpret_cast<const V8*>(resb + (size_t)rel2 * p.ldr);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
                *reinterpret_cast<f32x4*>(Cs + (lane & 15) * P2_CS_LD + j * 16 + (lane >> 4) * 4) = acc[j][i];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int rel = k == 0 ? 0 : (k == 1 ? 6 : rel2);
                const bool ok = act && (k < 2 || r0 < 4);
                const float* cs = Cs + (r0 + rel) * P2_CS_LD + ch * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(cs), hi = *reinterpret_cast<const f32x4*>(cs + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e] + fb0[e]; v[4 + e] = hi[e] + fb1[e]; }
                if constexpr (RES) {
                    const V8 res = fr[k];
                    if (i + 1 < MF) fr[k] = *reinterpret_cast<const V8*>(resb + (size_t)((i + 1) * 16 + rel) * p.ldr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += Tag::to_f32(res[e]);
                }
                V8 pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[e] = Tag::from_f32(v[e]);
                if (ok) *reinterpret_cast<V8*>(outb + (size_t)(i * 16 + rel) * p.ldc) = pk;
            }
        }
    };
    if (has_res) run(std::true_type()); else run(std::false_type());
}

// Section timing (development aid, mve_gemm_pp_profile): per wave, shader-clock sums of the four parts of a step.
__device__ unsigned long long* g_pp_prof = nullptr;

template <int V> struct PInt { static constexpr int value = V; };

// NSL: LDS ring slots (prefetch distance NSL - 1).  NSL = 3 (BN2 = 160, dense GEMM only): the tile that lets TWO blocks share a CU -- 128
// registers, 79 KiB -- so that one block's epilogue (memory-latency bound: residual reads, stores; no MFMA) runs under the other block's
// K loop and the fragment reads of either hide behind the other's MFMAs; its epilogue is per wave (pp2_epilogue), without block barriers.
template <class Tag, int MODE, bool SEQ, int BN2, bool PROF = false, int NSL = PNSLOT, bool PAIR = false, bool RED = false, bool LNF = false>        // MODE 1: slab-major (chunk64) conv only; PAIR: residual_pair epilogue; RED: in-kernel slice reduction (gemm_reduce_slices); LNF: LayerNorm of the output rows in the epilogue (big_tile_epilogue)
__global__ __launch_bounds__(PNTH, NSL == 3 ? 4 : 2) void k_gemm_pp(const GemmParams p) {
    typedef typename Tag::V8 V8;
    // BN2 = 320 | 256: waves 2 (M) x 4 (N), wave tile 128 x {80, 64};  BN2 = 128 (the 128-channel convolutions of the VAE at image
    // resolution): waves 4 x 2, wave tile 64 x 64 -- 16 MFMAs against 8 fragment reads per step, LDS-read bound (~3/4 of the MFMA rate)
    // BN2 = 160 (two tiles per 320 columns: small batches, where 256 x 320 tiles would leave CUs idle): waves 4 x 2, wave tile 64 x 80;
    // its 10 weight pieces per step go 3 / 3 / 2 / 2 to the group-0 waves, the last two add a zero-fill piece into a dump area so that
    // every wave of the group issues (and counts) three.
    constexpr int WAVES_N = (BN2 == 128 || BN2 == 160) ? 2 : 4, WAVES_M = 8 / WAVES_N;
    constexpr int WTM = PBM / WAVES_M, MF = WTM / 16;
    constexpr int WTN = BN2 / WAVES_N, NF = WTN / 16;
    constexpr int NPB_ALL = BN2 / 16, NPB_BASE = NPB_ALL / 4, NPB_REM = NPB_ALL % 4;
    constexpr int NPB = NPB_BASE + (NPB_REM ? 1 : 0);          // weight pieces per group-0 wave and step: 5 | 4 | 2 | 3 (160: 3, 3, 2 + filler, 2 + filler)
    constexpr int NPA = 4;                                     // activation pieces per group-1 wave and step
    constexpr int SLOT = pp_slot_bytes(BN2);
    constexpr int DIST = NSL - 1;                              // a step's pieces leave DIST steps before it is read
    static_assert(NSL == 4 || (NSL == 3 && BN2 == 160 && MODE == 0 && !SEQ), "three slots: the two-blocks-per-CU GEMM tile only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wid >> 2, widx = wid & 3;                 // wave group = DMA role; index of the wave inside its group
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;

    const int tiles_n = (p.N + BN2 - 1) / BN2;
    const int tiles_m = (p.M + PBM - 1) / PBM;
    const int S = p.splitk > 1 ? p.splitk : 1;
    const unsigned lin = mve_xcd_remap(blockIdx.x, (unsigned)(tiles_m * tiles_n * S));
    int kslice, tm, tn;
    if (p.w_major) {          // weight strip major (GemmParams::w_major): the row panels of one (column tile, K slice) strip are consecutive blocks
        tm = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)tiles_m));
        const unsigned rest = lin / (unsigned)tiles_m;
        kslice = __builtin_amdgcn_readfirstlane((int)(rest % (unsigned)S));
        tn = __builtin_amdgcn_readfirstlane((int)(rest / (unsigned)S));
    } else {
        kslice = lin % S;
        const unsigned t_ = lin / S;
        tm = t_ / tiles_n; tn = t_ % tiles_n;
    }
    const unsigned tile = (unsigned)(tm * tiles_n + tn);
    const int m0 = tm * PBM, n0 = tn * BN2;

    const int nk_all = p.K / BK;
    // (the 64-bit divisions are expanded on the VALU: readfirstlane returns the results to SGPRs)
    const int kt_begin = __builtin_amdgcn_readfirstlane((int)((long long)nk_all * kslice / S));
    const int kt_end = __builtin_amdgcn_readfirstlane((int)((long long)nk_all * (kslice + 1) / S));
    const int nsteps = 2 * (kt_end - kt_begin);

    f32x4 acc[NF][MF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long prof_v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_start = 0, prof_loop0 = 0, prof_end_loop = 0;
    if constexpr (PROF) prof_start = __builtin_readcyclecounter();
    // Everything from here to the end of the K loop exists once per role (ROLE 0: waves 0-3 stream the weight rows, ROLE 1: waves 4-7
    // the activation rows): two straight-line copies instead of role tests, and their loop-carried copies, inside every step.
    // (each role's copy of the loop starts its own accumulators: defined once in front of the role branch, hipcc parks 23 of them in scratch around the loops)
    auto run = [&](auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
    if constexpr (PAIR) {
        // residual_pair mode: the accumulators START from the residual, hi + lo (exact in fp32), read in the accumulator layout -- a lane owns 4
        // consecutive columns of a row per fragment: 8-byte loads, 16 rows x 32 B per instruction -- before any LDS-DMA is in flight (the loads are
        // the compiler's: it waits for them where it converts them).  The epilogue then has no load behind its stores and no residual to prefetch
        // next to 160 accumulators (gemm_big_epilogue.h).  A split-K slice starts from zero: the reducer adds the residual.  Rows past M are clamped
        // (never stored).  The sum is residual + sum_k a w, then + bias: the same value as the other kernels' (sum + bias) + residual up to the order
        // of two fp32 additions.
        if (p.residual && p.splitk <= 1) {
            typedef typename Tag::T T;
            typedef T T4 __attribute__((ext_vector_type(4)));
            const T* rh = reinterpret_cast<const T*>(p.residual);
            const unsigned char* rl = reinterpret_cast<const unsigned char*>(p.residual_lo);      // lo8: one byte per element (common.h)
            const int col = n0 + wn * WTN + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < MF; ++i) {                   // one fragment row at a time (10 loads in flight): hoisted to the top, the 80 loads would spill
                int m = m0 + wm * WTM + i * 16 + (lane & 15);
                m = m < p.M ? m : p.M - 1;
                const size_t ro = (size_t)m * p.ldr + col;
                T4 h[NF];
                unsigned l[NF];
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    h[j] = *reinterpret_cast<const T4*>(rh + ro + j * 16);
                    l[j] = rl ? *reinterpret_cast<const unsigned*>(rl + ro + j * 16) : 0u;       // (E5M2 zero is all-zero bits)
                }
#pragma unroll
                for (int j = 0; j < NF; ++j) {               // (whole-vector assignment: element writes through the captured reference keep `acc` in scratch)
                    f32x4 a;
                    float lf[4];
                    mve_lo8_unpack4(l[j], lf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = Tag::to_f32(h[j][e]) + lf[e];
                    acc[j][i] = a;
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of the compiler's may be in flight when the counted LDS-DMA waits begin
        }
    }
    constexpr int NPM = ROLE == 0 ? NPB : NPA;                 // pieces this wave issues per step
    // ---- DMA role state ------------------------------------------------------------------------------------------------------
    // lane l of a piece writes LDS bytes [16 l, 16 l + 16): row l >> 2 of the piece, stored chunk l & 3 = source chunk ^ swizzle
    const int prow = lane >> 2;
    // row of the piece = lane >> 2: swizzle (row >> 1) & 3  (p.old_swizzle: the round-2 term (row >> 2) & 3, 2-way conflicted -- same-box A/B only)
    const unsigned c16 = (unsigned)(((lane & 3) ^ ((lane >> (p.old_swizzle ? 4 : 3)) & 3)) << 4);
    unsigned vo[5] = {0u, 0u, 0u, 0u, 0u};    // per-piece byte offset of the lane's chunk: constant for weights / dense A, per (tap, slab) for conv rows
    unsigned pixb[NPA] = {0u, 0u, 0u, 0u};    // conv rows: biased pixel index of the window origin
    unsigned vmask[NPA] = {0u, 0u, 0u, 0u};   // conv rows: 9-bit tap validity + window-origin parities (upsample)
    const unsigned smem_base = (unsigned)(size_t)smem;
    unsigned long long op_base;               // group 0: weights; group 1: dense A (MODE 0)
    unsigned dst0;                            // LDS offset of this wave's first piece inside a slot
    const int conv_bias = p.g.Ws + 1;
    const bool win2 = p.g.kw == 2;            // 2 x 2 window (a phase of upsample + conv) instead of the 3 x 3 taps: uniform
    const int KW = win2 ? 2 : 3, KT = win2 ? 4 : 9;
    if constexpr (ROLE == 0) {
        const int pstart = widx * NPB_BASE + (widx < NPB_REM ? widx : NPB_REM);      // first piece of this wave
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            int n = n0 + (pstart + q) * 16 + prow;
            n = n < p.N ? n : p.N - 1;
            vo[q] = (unsigned)n * (unsigned)p.ldw * 2u + c16;
        }
        if (NPB_REM && widx >= NPB_REM) vo[NPB - 1] = 0xFFFFFFFFu;                    // filler piece: out of range, zeros
        op_base = (unsigned long long)p.W + (unsigned long long)kt_begin * (BK * 2);
        if constexpr (MODE == 1) {             // all four phases in one launch: this tile's phase picks its weight block
            if (p.g.phase_rows > 0) op_base += (unsigned long long)(m0 / p.g.phase_rows) * ((unsigned long long)p.N * p.ldw * 2ull);
        }
        dst0 = smem_base + P_A_SLOT + pstart * 1024;
    } else {
        const int hw = p.g.Ho * p.g.Wo;
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            int m = m0 + (widx * NPA + q) * 16 + prow;
            m = m < p.M ? m : p.M - 1;
            if constexpr (MODE == 0) vo[q] = (unsigned)m * (unsigned)p.lda * 2u + c16;
            else {
                int pad_y = p.g.pad, pad_x = p.g.pad_x;
                if (p.g.phase_rows > 0) {              // rows [ph R, (ph + 1) R): phase ph of the upsampler
                    const int ph = conv_phase_of(m, p.g.phase_rows);
                    m -= ph * p.g.phase_rows;
                    pad_y = 1 - (ph >> 1); pad_x = 1 - (ph & 1);
                }
                const int b = m / hw, r = m - b * hw;
                const int y = r / p.g.Wo;
                const int cy = y * p.g.stride - pad_y, cx = (r - y * p.g.Wo) * p.g.stride - pad_x;
                pixb[q] = (unsigned)((b * p.g.Hs + (cy >> p.g.ups)) * p.g.Ws + (cx >> p.g.ups) + conv_bias);
                unsigned mk = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yi = cy + (win2 ? t >> 1 : t / 3), xi = cx + (win2 ? t & 1 : t % 3);
                    if (t < KT && yi >= 0 && yi < p.g.Hv && xi >= 0 && xi < p.g.Wv) mk |= 1u << t;
                }
                if (p.g.ups) mk |= ((unsigned)(cy & 1) << 9) | ((unsigned)(cx & 1) << 10);
                vmask[q] = mk;
            }
        }
        op_base = (unsigned long long)p.A + (unsigned long long)kt_begin * (BK * 2);
        dst0 = smem_base + widx * NPA * 1024;
    }
    unsigned long long conv_base = 0;         // resource base of the current (tap, slab), first 32-channel half
    // conv sources as opaque scalars: left to itself hipcc turns `second ? p.g.C2 : p.g.C1` into a select of kernarg ADDRESSES and an
    // s_load + lgkmcnt(0) inside the K loop
    const int gC1 = p.g.C1, gC3 = p.g.C3, g_nkm = p.g.nk_main;
    const int dC21 = p.g.C2 - p.g.C1, dC43 = p.g.C4 - p.g.C3;             // (differences: arithmetic results cannot be re-materialised as loads)
    const long long gA1 = (long long)p.A, gA3 = (long long)p.A3;
    const long long dA21 = (long long)p.A2 - (long long)p.A, dA43 = (long long)p.A4 - (long long)p.A3;
    int i_kt = kt_begin, i_tap = kt_begin % KT, i_c0 = (kt_begin / KT) * 64, cs_cur = -1;      // issue-side K position (conv)
    int d_px = 0, d_row = 0;                  // byte step of the base from one tap to the next: same row / next row
    unsigned vrow[NPA] = {0u, 0u, 0u, 0u};    // conv rows: byte offset of the lane's chunk at the window origin for the current source
    i32x4 r_cur;                              // resource and LDS destination of the step being issued
    unsigned dst_cur;
    unsigned dst_last = 0;                    // destination of piece NPB - 1: the dump area for a filler piece
    int slot_is = 0;                          // ring slot of the step being issued (steps are issued in order)

    // address part of issuing step s (relative to kt_begin): scalar work, plus 4 x ~5 VALU per 64-channel slab on the conv rows.
    // Steps past the end get a zero-sized resource: their pieces write zeros and keep the vmcnt bookkeeping uniform.
    auto prep = [&](int s) {
        const bool live = s < nsteps;
        dst_cur = dst0 + (unsigned)slot_is * SLOT;
        slot_is = slot_is + 1 == NSL ? 0 : slot_is + 1;
        if constexpr (ROLE == 0 && NPB_REM != 0) dst_last = widx >= NPB_REM ? smem_base + NSL * SLOT : dst_cur + (NPB - 1) * 1024;
        if constexpr (MODE == 0 || ROLE == 0) {
            r_cur = pp_rsrc(op_base + (unsigned long long)s * PROWB, live);
        } else {
            if (!(s & 1)) {
                // (tap, slab) of the K tile being issued (issue order is sequential: the counters advance by one tile per even step).
                // Inside a 64-channel slab of the 3x3 part only the tap moves: the base advances by one pixel, or by a row minus two
                // pixels, of the current source; everything else (slab, source, shortcut part) takes the full decode.
                const bool shortcut = g_nkm > 0 && i_kt >= g_nkm;
                int t_ = i_tap;
                if (i_tap == 0 || shortcut || p.g.ups || cs_cur < 0) {       // (cs_cur < 0: first tile of a split-K slice that starts inside a slab)
                    int c0 = i_c0;
                    bool second = c0 >= gC1;
                    long long src = gA1 + (second ? dA21 : 0ll);
                    int cs = gC1 + (second ? dC21 : 0);
                    int ch = second ? c0 - gC1 : c0;
                    if (shortcut) {                                    // 1x1 shortcut part: centre tap of the shortcut sources
                        t_ = 4;
                        c0 = (i_kt - g_nkm) * 64;
                        second = c0 >= gC3;
                        src = gA3 + (second ? dA43 : 0ll);
                        cs = gC3 + (second ? dC43 : 0);
                        ch = second ? c0 - gC3 : c0;
                    }
                    const int dy = win2 ? t_ >> 1 : (t_ * 11) >> 5, dx = t_ - dy * KW;    // (t_ * 11) >> 5 = t_ / 3 for 0 <= t_ < 9
                    const int toff = p.g.ups ? 0 : dy * p.g.Ws + dx;
                    conv_base = (unsigned long long)(src + 2ll * ((long long)(toff - conv_bias) * cs + ch));
                    d_px = p.g.ups ? 0 : cs * 2;
                    d_row = p.g.ups ? 0 : (p.g.Ws - (KW - 1)) * cs * 2;
                    if (cs != cs_cur || p.g.ups) {                     // the row offsets depend on the source's channel count only (and on the tap when upsampling)
                        cs_cur = cs;
#pragma unroll
                        for (int q = 0; q < NPA; ++q) {
                            unsigned px = pixb[q];
                            if (p.g.ups) px += (unsigned)((int)((((vmask[q] >> 9) & 1u) + dy) >> 1) * p.g.Ws + (int)((((vmask[q] >> 10) & 1u) + dx) >> 1));
                            vrow[q] = px * ((unsigned)cs * 2u) + c16;
                        }
                    }
                } else {
                    conv_base += (unsigned long long)(long long)((win2 ? i_tap == 2 : (i_tap == 3 || i_tap == 6)) ? d_row : d_px);
                }
                ++i_kt;
                if (++i_tap == KT) { i_tap = 0; i_c0 += 64; }
#pragma unroll
                for (int q = 0; q < NPA; ++q)      // halo row: all-ones offset = out of range (two VALU: v_bfe_i32 of the inverted mask, v_or)
                    vo[q] = vrow[q] | (unsigned)__builtin_amdgcn_sbfe((int)~vmask[q], (unsigned)t_, 1u);
            }
            r_cur = pp_rsrc(conv_base + ((s & 1) ? PROWB : 0), live);
        }
    };
    // (q is a compile-time constant at every call site; group 1 has no fifth piece)
    auto piece_dst = [&](int q) { return (ROLE == 0 && NPB_REM != 0 && q == NPB - 1) ? dst_last : dst_cur + q * 1024; };
    auto piece_aim = [&](int q) { if (q < NPM) pp_set_m0(piece_dst(q)); };
    auto piece_fire = [&](int q) { if (q < NPM) pp_dma_m0(vo[q], r_cur); };
    auto piece_now = [&](int q) { if (q < NPM) pp_piece_now(vo[q], r_cur, piece_dst(q)); };

    // fragment read offsets inside a slot (lane constants): row r = ... + frow, stored chunk = fchunk ^ ((r >> 1) & 3)
    const int frow = lane & 15, fchunk = lane >> 4;
    const int fsw = (fchunk ^ ((frow >> (p.old_swizzle ? 2 : 1)) & 3)) << 4;
    const int a_off = (wm * WTM + frow) * PROWB + fsw;
    const int b_off = P_A_SLOT + (wn * WTN + frow) * PROWB + fsw;

    // this wave's pieces of all but the newest DIST - 1 / DIST - 2 issued steps have landed (four slots: 2 / 1; three slots: 1 / 0)
    auto wait_keep2 = [&]() { pp_wait_vm<(DIST - 1) * NPM>(); };
    auto wait_keep1 = [&]() { pp_wait_vm<(DIST - 2) * NPM>(); };

    if constexpr (PROF) prof_loop0 = __builtin_readcyclecounter();
    const unsigned m0_keep = pp_m0_take();
#pragma unroll 1
    for (int s = 0; s < DIST; ++s) {
        prep(s);
#pragma unroll
        for (int q = 0; q < 5; ++q) piece_now(q);
    }
    wait_keep2();                             // step 0
    pp_barrier();
    if constexpr (ROLE == 1) pp_barrier();    // group 1 starts one interval late

    const int SQ = SEQ ? p.splitk_seq : 1;
    int sl_idx = 0;
    int fold_at = SEQ ? 2 * __builtin_amdgcn_readfirstlane((int)((long long)nk_all * 1 / SQ)) : nsteps;      // first step index at which a slice is complete

    // One copy of the step body: the slot offset is a run-time scalar (two v_add per step) rather than four unrolled copies --
    // the SEQ fold below would otherwise be inlined into each of them.
    unsigned long long t_l = 0, t_b1 = 0, t_m = 0, t_b2 = 0, t_rd = 0, t_prep = 0;
    int slot_rd = 0;                          // ring slot of the step being read
    auto step = [&](int k) {
        V8 xf[MF], wf[NF];
        unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, ca = 0, cb = 0;
        if constexpr (PROF) c0 = __builtin_readcyclecounter();
        const unsigned char* sp = smem + slot_rd * SLOT;
        slot_rd = slot_rd + 1 == NSL ? 0 : slot_rd + 1;
        // ---- L(k): fragments of step k; addresses of step k + 3; this wave's pieces of step k + 1 have landed ----
        if constexpr (NSL == 3) {
            // three slots: the pieces of step k + 2 leave at the TOP of L(k) (their slot was last read in L(k-1), one barrier ago for either group),
            // two intervals before their wait at the end of L(k+1): issued from M(k) they had one interval, less than an HBM round trip
            prep(k + DIST);
#pragma unroll
            for (int q = 0; q < 5; ++q) piece_now(q);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) wf[j] = *reinterpret_cast<const V8*>(sp + b_off + j * 1024);
#pragma unroll
        for (int i = 0; i < MF; ++i) xf[i] = *reinterpret_cast<const V8*>(sp + a_off + i * 1024);
        if constexpr (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ca = __builtin_readcyclecounter(); }
        if constexpr (NSL != 3) prep(k + DIST);
        if constexpr (PROF) cb = __builtin_readcyclecounter();
        if constexpr (NSL == 3) pp_wait_vm<NPM>();          // in flight: steps k + 1 (issued in L(k-1)) and k + 2 (just now) -> k + 2 may stay
        else wait_keep1();                         // four slots: in flight are steps k + 1 (issued in M(k-2)) and k + 2 (M(k-1)) -> k + 2 may stay; three: k + 1 only
        if constexpr (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c1 = __builtin_readcyclecounter(); }
        pp_barrier_lds();
        if constexpr (PROF) c2 = __builtin_readcyclecounter();
        // ---- M(k): MF MFMAs per weight fragment; one LDS-DMA piece of step k + DIST goes out behind each group, its M0 one MFMA earlier ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                PMfma<Tag>::run(acc[j][i], wf[j], xf[i]);
                if constexpr (NSL != 3) { if (i == MF - 2) piece_aim(j); }
            }
            if constexpr (NSL != 3) piece_fire(j);
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (PROF) c3 = __builtin_readcyclecounter();
        pp_barrier();
        if constexpr (PROF) {
            const unsigned long long c4 = __builtin_readcyclecounter();
            t_l += c1 - c0; t_b1 += c2 - c1; t_m += c3 - c2; t_b2 += c4 - c3; t_rd += ca - c0; t_prep += cb - ca;
        }
        if constexpr (SEQ) {
            // sequential split-K emulation (GemmParams::splitk_seq): at a slice boundary the accumulators are folded into lane-private
            // 16-byte slots of a block-private fp32 running total.  The fold's loads / stores are the compiler's: drain the DMA queue
            // around it so that its vmcnt bookkeeping and the counted waits above stay exact.
            if (k + 1 == fold_at) {
                pp_mfma_settle();
                pp_wait_vm<0>();
                asm volatile("; pp_fold_begin");          // tools/check_pp_isa.py: compiler-generated memory traffic is legal only up to pp_fold_end
                // (the offset passes through an opaque asm so that the 40 slot addresses are formed here, not hoisted out of the K loop
                // into 80 registers that would be spilled -- scratch reloads on the hot path would drain vmcnt)
                size_t tot_off = (size_t)tile * (PBM * BN2 / 4) + (size_t)wid * (NF * MF * 64) + lane;
                asm volatile("" : "+v"(tot_off));
                f32x4* tot = reinterpret_cast<f32x4*>(p.partial) + tot_off;
                const bool last = sl_idx + 1 == SQ;
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int i = 0; i < MF; ++i) {
                        f32x4* slot = tot + (j * MF + i) * 64;
                        f32x4 t = acc[j][i];
                        if (sl_idx > 0) { const f32x4 o = *slot; t = f32x4{o[0] + t[0], o[1] + t[1], o[2] + t[2], o[3] + t[3]}; }
                        if (last) acc[j][i] = t;
                        else { *slot = t; acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                asm volatile("; pp_fold_end");
                pp_wait_vm<0>();
                ++sl_idx;
                fold_at = 2 * __builtin_amdgcn_readfirstlane((int)((long long)nk_all * (sl_idx + 1) / SQ));
            }
        }
    };

#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) step(k);
    pp_m0_give(m0_keep);
    if constexpr (ROLE == 0) pp_barrier();    // pairs with group 1's last barrier
    if constexpr (PROF) {
        prof_v[0] = t_l; prof_v[1] = t_b1; prof_v[2] = t_m; prof_v[3] = t_b2; prof_v[4] = t_rd; prof_v[5] = t_prep;
        prof_v[6] = prof_loop0 - prof_start;
        prof_end_loop = __builtin_readcyclecounter();
    }
    };  // run
    if (role == 0) run(PInt<0>());
    else run(PInt<1>());
    pp_wait_vm<0>();                          // the zero-fill pieces of the steps past the end
    pp_barrier();
    pp_mfma_settle();
    if constexpr (NSL == 3) pp2_epilogue<Tag>(p, acc, smem, m0, n0, wid, lane, wm, wn);
    else big_tile_epilogue<Tag, BN2, WAVES_N, PAIR, LNF>(p, acc, smem, m0, n0, kslice, tid, lane, wm, wn);
    if constexpr (RED) gemm_reduce_slices<Tag, BN2, PBM, PNTH>(p, tile, kslice, S, m0, n0, tid);
    if constexpr (PROF) {
        if (g_pp_prof && lane == 0) {
            __builtin_amdgcn_s_waitcnt(0);     // the epilogue's stores have left the wave (vmcnt / lgkmcnt / expcnt all zero)
            prof_v[7] = __builtin_readcyclecounter() - prof_end_loop;
            unsigned long long* o = g_pp_prof + ((size_t)blockIdx.x * 8 + wid) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = prof_v[i];
        }
    }
}

// every per-lane byte offset must stay below the resource size (and the out-of-range marker above it)
bool pp_fits(unsigned long long bytes) { return bytes + 65536ull < (unsigned long long)P_NUMREC; }

int pp_bn(int N) { return N % 320 == 0 ? 320 : (N % 256 == 0 ? 256 : (N % 128 == 0 ? 128 : 0)); }
int pp_bn(const GemmParams& p) { return (p.tile_n == 160 && p.N % 160 == 0) ? 160 : pp_bn(p.N); }

unsigned long long* g_pp_prof_host = nullptr;

// set by a launch that normalised its output rows itself (GemmParams::ln_out); read and cleared by mve_gemm_pp_ln_fused()
thread_local bool g_pp_ln_fused = false;
int g_pp_ln_fuse = -1;      // MVE_GEMM_LN_FUSE (default 1); 0: never (A/B, tests)
bool pp_ln_fusable(const GemmParams& p) {
    if (g_pp_ln_fuse < 0) { const char* e = getenv("MVE_GEMM_LN_FUSE"); g_pp_ln_fuse = e ? atoi(e) : 1; }
    return g_pp_ln_fuse != 0 && p.N == 320 && p.M % PBM == 0 && p.out_lo && p.bias && p.out_scale == 1.0f && !p.res_after_scale && !p.out_f32 && !p.geglu &&
           !p.rowvec && p.splitk <= 1 && p.splitk_seq <= 1 && p.orow_extra == 0 && p.tile_n == 0 && !(p.dbg & 2) && p.ln_gamma && p.ln_beta && p.ld_ln % 8 == 0;
}

template <class Tag, int MODE, bool SEQ, int BN2>
int launch_pp3(const GemmParams& p, hipStream_t s) {
    if (p.sk_sync && p.splitk > 1) {          // K slices folded inside the launch (the partial tiles leave raw: the plain instantiation's generic epilogue path)
        if constexpr (!SEQ && (BN2 == 320 || BN2 == 160)) {
            static bool configured_red[64] = {};
            int dev = 0;
            MVE_HIP(hipGetDevice(&dev));
            if (dev >= 0 && dev < 64 && !configured_red[dev]) {
                MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pp<Tag, MODE, SEQ, BN2, false, PNSLOT, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem(BN2)));
                configured_red[dev] = true;
            }
            const unsigned grid = (unsigned)mve_cdiv(p.M, PBM) * (unsigned)mve_cdiv(p.N, BN2) * (unsigned)p.splitk;
            k_gemm_pp<Tag, MODE, SEQ, BN2, false, PNSLOT, false, true><<<grid, PNTH, pp_smem(BN2), s>>>(p);
            MVE_LAUNCH_CHECK();
            return MVE_OK;
        } else {
            return 1;
        }
    }
    if constexpr (!SEQ && BN2 == 320 && MODE == 0) {
        // LayerNorm of the output rows inside the launch: exactly the configuration big_tile_epilogue's pair fast path takes on EVERY tile of the launch
        if (p.ln_out && pp_ln_fusable(p)) {
            static bool configured_ln[64] = {};
            int dev = 0;
            MVE_HIP(hipGetDevice(&dev));
            if (dev >= 0 && dev < 64 && !configured_ln[dev]) {
                MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pp<Tag, 0, false, 320, false, PNSLOT, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem(320)));
                configured_ln[dev] = true;
            }
            const unsigned grid = (unsigned)(p.M / PBM);
            k_gemm_pp<Tag, 0, false, 320, false, PNSLOT, true, false, true><<<grid, PNTH, pp_smem(320), s>>>(p);
            MVE_LAUNCH_CHECK();
            g_pp_ln_fused = true;
            return MVE_OK;
        }
    }
    if (p.residual_lo || p.out_lo) {          // residual_pair mode: the 320-wide and (round 5: small batches) 160-wide tiles without the slice fold
        if constexpr (!SEQ && (BN2 == 320 || BN2 == 160)) {
            static bool configured_pair[64] = {};
            int dev = 0;
            MVE_HIP(hipGetDevice(&dev));
            if (dev >= 0 && dev < 64 && !configured_pair[dev]) {
                MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pp<Tag, MODE, SEQ, BN2, false, PNSLOT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem(BN2)));
                configured_pair[dev] = true;
            }
            const unsigned grid = (unsigned)mve_cdiv(p.M, PBM) * (unsigned)mve_cdiv(p.N, BN2) * (unsigned)(p.splitk > 1 ? p.splitk : 1);
            k_gemm_pp<Tag, MODE, SEQ, BN2, false, PNSLOT, true><<<grid, PNTH, pp_smem(BN2), s>>>(p);
            MVE_LAUNCH_CHECK();
            return MVE_OK;
        } else {
            return 1;                         // not eligible: the caller falls back (128-row kernel: gemm_epilogue_tail carries the pair)
        }
    }
    if constexpr (std::is_same<Tag, F16Tag>::value && !SEQ && BN2 == 320) {
        if (g_pp_prof_host) {
            MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pp<Tag, MODE, SEQ, BN2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem(BN2)));
            const unsigned grid = (unsigned)mve_cdiv(p.M, PBM) * (unsigned)mve_cdiv(p.N, BN2) * (unsigned)(p.splitk > 1 ? p.splitk : 1);
            k_gemm_pp<Tag, MODE, SEQ, BN2, true><<<grid, PNTH, pp_smem(BN2), s>>>(p);
            MVE_LAUNCH_CHECK();
            return MVE_OK;
        }
    }
    static bool configured[64] = {};
    int dev = 0;
    MVE_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !configured[dev]) {
        MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pp<Tag, MODE, SEQ, BN2>), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem(BN2)));
        configured[dev] = true;
    }
    const unsigned grid = (unsigned)mve_cdiv(p.M, PBM) * (unsigned)mve_cdiv(p.N, BN2) * (unsigned)(p.splitk > 1 ? p.splitk : 1);
    k_gemm_pp<Tag, MODE, SEQ, BN2><<<grid, PNTH, pp_smem(BN2), s>>>(p);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

// The two-blocks-per-CU tile: dense GEMM, whole 256 x 160 tiles, the epilogue configuration pp2_epilogue implements
bool pp2_eligible(int mode, const GemmParams& p) {
    if (mode != 0 || p.N % 160 != 0 || p.M % PBM != 0 || p.K % BK != 0) return false;
    if (p.splitk > 1 || p.splitk_seq > 1 || p.rowvec || p.out_f32 || p.out_scale != 1.0f || (p.residual && p.res_after_scale)) return false;
    if (p.residual_lo || p.out_lo) return false;          // the pair epilogue lives in big_tile_epilogue / gemm_epilogue_tail only
    if (p.geglu && (p.ldc % 8 != 0)) return false;
    if (!p.geglu && p.ldc % 8 != 0) return false;
    return pp_fits((unsigned long long)p.N * p.ldw * 2) && pp_fits((unsigned long long)p.M * p.lda * 2);
}

template <class Tag>
int launch_pp2(const GemmParams& p, hipStream_t s) {
    static bool configured[64] = {};
    int dev = 0;
    MVE_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !configured[dev]) {
        MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pp<Tag, 0, false, 160, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem(160, 3)));
        configured[dev] = true;
        if (getenv("MVE_DEBUG")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&k_gemm_pp<Tag, 0, false, 160, false, 3>), PNTH, pp_smem(160, 3));
            fprintf(stderr, "[mve] k_gemm_pp<160, 3 slots>: %d blocks per CU by the occupancy API (LDS %d B per block)\n", nb, pp_smem(160, 3));
        }
    }
    const unsigned grid = (unsigned)(p.M / PBM) * (unsigned)(p.N / 160);
    if constexpr (std::is_same<Tag, F16Tag>::value) {
        if (g_pp_prof_host) {
            MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pp<Tag, 0, false, 160, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, pp_smem(160, 3)));
            k_gemm_pp<Tag, 0, false, 160, true, 3><<<grid, PNTH, pp_smem(160, 3), s>>>(p);
            MVE_LAUNCH_CHECK();
            return MVE_OK;
        }
    }
    k_gemm_pp<Tag, 0, false, 160, false, 3><<<grid, PNTH, pp_smem(160, 3), s>>>(p);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

template <class Tag, int MODE>
int launch_pp(const GemmParams& p, hipStream_t s) {
    if (pp_bn(p) == 256) return launch_pp3<Tag, MODE, false, 256>(p, s);
    if (pp_bn(p) == 128) return launch_pp3<Tag, MODE, false, 128>(p, s);
    if (pp_bn(p) == 160) return launch_pp3<Tag, MODE, false, 160>(p, s);
    return p.splitk_seq > 1 ? launch_pp3<Tag, MODE, true, 320>(p, s) : launch_pp3<Tag, MODE, false, 320>(p, s);
}

int g_pp_old_swizzle = 0;

bool pp_eligible(int mode, const GemmParams& p) {
    const int bn = pp_bn(p);
    if (bn == 0 || p.M < 64 || p.K % BK != 0) return false;
    if (bn != 320 && bn != 160 && p.splitk > 1) return false;
    if (bn != 320 && p.splitk_seq > 1) return false;
    if (!pp_fits((unsigned long long)p.N * p.ldw * 2)) return false;
    if (mode == 0) return pp_fits((unsigned long long)p.M * p.lda * 2);
    if (!p.g.chunk64) return false;
    const int hw = p.g.Ho * p.g.Wo;
    if (p.g.phase_rows > 0 && (p.g.kw != 2 || p.g.phase_rows % PBM != 0 || p.M != 4 * p.g.phase_rows)) return false;
    const int m_img = p.g.phase_rows > 0 ? p.g.phase_rows : p.M;      // rows that address distinct source pixels
    const unsigned long long px = (unsigned long long)((m_img + hw - 1) / hw) * p.g.Hs * p.g.Ws + 2ull * p.g.Ws + 4;
    int cmax = p.g.C1 > p.g.C2 ? p.g.C1 : p.g.C2;
    cmax = cmax > p.g.C3 ? cmax : p.g.C3;
    cmax = cmax > p.g.C4 ? cmax : p.g.C4;
    return pp_fits(px * cmax * 2);
}

}  // namespace

// development aid: with a device buffer of 32 x (number of blocks) uint64 set, fp16 320-wide launches run the instrumented kernel and
// every wave writes its shader-clock sums {L work, wait at the L barrier, M work, wait at the M barrier}; nullptr turns it off
extern "C" MVE_API int mve_gemm_pp_profile(void* buf) {
    g_pp_prof_host = reinterpret_cast<unsigned long long*>(buf);
    MVE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pp_prof), &g_pp_prof_host, sizeof(g_pp_prof_host)));
    return MVE_OK;
}

// did the last launch of this thread normalise its output rows itself?  (read once: the flag is cleared)
bool mve_gemm_pp_ln_fused() {
    const bool f = g_pp_ln_fused;
    g_pp_ln_fused = false;
    return f;
}
int mve_gemm_pp_ln_fuse_tune(int on) {       // -> previous setting (the environment default resolved); negative only queries
    if (g_pp_ln_fuse < 0) { const char* e = getenv("MVE_GEMM_LN_FUSE"); g_pp_ln_fuse = e ? atoi(e) : 1; }
    const int old = g_pp_ln_fuse != 0;
    if (on >= 0) g_pp_ln_fuse = on ? 1 : 0;
    return old;
}

// A/B aid: 1 = the ring swizzle of round 2 ((row >> 2) & 3: every fragment read 2-way bank conflicted); results are identical either way
void mve_gemm_pp_old_swizzle(int on) { g_pp_old_swizzle = on ? 1 : 0; }

// ping-pong main loop + epilogue (or split-K partials; the caller runs the reducer).  Returns 1 when the problem is not eligible
// (the caller falls back to the two-stage kernel), MVE_OK after a launch, < 0 on error.
int mve_gemm_pp_launch(int dtype, int mode, const void* params, void* stream) {
    GemmParams p = *reinterpret_cast<const GemmParams*>(params);
    p.old_swizzle = g_pp_old_swizzle;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("MVE_PP_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    hipStream_t s = (hipStream_t)stream;
    if (p.tile_n == 161) {               // the caller asks for the two-blocks-per-CU tile
        if (!pp2_eligible(mode, p)) return 1;
        if (dtype == MVE_F16) return launch_pp2<F16Tag>(p, s);
        if (dtype == MVE_BF16) return launch_pp2<BF16Tag>(p, s);
        mve_set_error("gemm_pp: unsupported dtype %d", dtype);
        return MVE_ERR_ARG;
    }
    if (!pp_eligible(mode, p)) return 1;
    if (dtype == MVE_F16) return mode == 0 ? launch_pp<F16Tag, 0>(p, s) : launch_pp<F16Tag, 1>(p, s);
    if (dtype == MVE_BF16) return mode == 0 ? launch_pp<BF16Tag, 0>(p, s) : launch_pp<BF16Tag, 1>(p, s);
    mve_set_error("gemm_pp: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}
