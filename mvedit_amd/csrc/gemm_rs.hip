// Role-split MFMA GEMM / implicit-GEMM conv for gfx950 (variant 2 of mve_gemm / mve_conv3x3).
//
// Why: rocprofv3 on the 128-row kernel (gemm.hip) showed the matrix pipe busy only ~36 % of the time: every
// wave alternates "issue ds_reads, wait for LDS, issue MFMAs", and the two waves that share a SIMD (two
// independent blocks) drift into the same phase.  Here the anti-phase is made structural:
//
//   * one block = 512 threads = 8 waves = two GROUPS of 4 waves; group g owns rows [128g, 128g+128) of a
//     256 x BN output tile; both groups share the W tile (half the W traffic per flop);
//   * each K tile (BK = 64) is processed in two phases separated by raw s_barriers:
//         phase L: issue the LDS-DMA for K tile kt+2, read ALL fragments of tile kt from LDS into registers,
//                  wait (counted vmcnt) for this wave's DMA of tile kt+1;
//         phase M: 32..40 back-to-back MFMAs on registers only (s_setprio 1);
//     group 1 executes ONE extra barrier before its loop, so it is always one phase behind group 0: on every SIMD one
//     wave is in phase M while its partner is in phase L, and the matrix pipe always has a wave feeding it;
//   * LDS: 3 stages x (A0 16 KB + A1 16 KB + W BN*128 B) = 156 KB at BN = 160; one block per CU.
//
// Hazards (phi = global phase index = number of barriers passed; group 0 runs L at even phi, group 1 at odd phi):
//   RAW  DMA for tile kt+1 is issued in L(kt-1) and waited for (vmcnt(#DMA of this phase)) at the end of L(kt) by the
//        issuing wave, i.e. before the barrier that ends phi = 2kt (group 0) / 2kt+1 (group 1).  First readers:
//        group 0 in L(kt+1) at phi = 2kt+2, after both barriers.
//   WAR  stage kt%3 is read in L(kt) (phi = 2kt / 2kt+1; reads drained with lgkmcnt(0) before the barrier) and
//        overwritten by the DMA for tile kt+3, issued in L(kt+1) at phi = 2kt+2 (group 0: A0 + its W rows) and
//        phi = 2kt+3 (group 1) -- both after the last reader's barrier.
// Epilogue and operand conventions are those of gemm.hip (swapped MFMA, fp32 LDS-staged epilogue).
#include "gemm_shared.h"

namespace {

constexpr int NT2 = 512;
constexpr int NSTAGE = 3;

template <class Tag, int BN, int MODE>
__global__ __launch_bounds__(NT2, 2) void k_gemm_rs(const GemmParams p) {
    constexpr int WN = BN / 2;
    constexpr int NF = WN / 16;
    constexpr int MF = 4;
    constexpr int A_STAGE = 128 * ROW_BYTES;            // per group
    constexpr int W_STAGE = BN * ROW_BYTES;
    constexpr int STAGE = 2 * A_STAGE + W_STAGE;
    constexpr int W_PASSES = BN / 32;                   // 32-row passes of the W tile (5 / 4 / 2)
    constexpr int WP0 = (W_PASSES + 1) / 2;             // passes issued by group 0 (even ones)
    constexpr int WP1 = W_PASSES / 2;                   // passes issued by group 1 (odd ones)
    constexpr int NDMA0 = 4 + WP0, NDMA1 = 4 + WP1;     // LDS-DMA instructions per wave per K tile
    constexpr int CS_LD = BN + 4;
    constexpr int CS_BYTES = 64 * CS_LD * 4;            // fp32 epilogue staging per group per pass
    static_assert(2 * CS_BYTES <= NSTAGE * STAGE, "epilogue staging must fit in the pipeline LDS");
    typedef typename Tag::V8 V8;
    typedef typename Tag::T T;

    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);     // wave-uniform group id
    const int gt = tid & 255;                                      // thread index inside the group
    const int gw = gt >> 6;                                        // wave inside the group
    const int wm = gw >> 1, wn = gw & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + 255) / 256;
    const unsigned tile = mve_xcd_remap(blockIdx.x, (unsigned)(tiles_m * tiles_n));
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * 256 + grp * 128, n0 = tn * BN;

    const int lr = gt >> 3;
    const int lc = (gt & 7) ^ ((lr >> 1) & 7);          // logical chunk fetched into physical slot gt&7 (swizzle on the source)
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.W);
    const T* zero = reinterpret_cast<const T*>(g_zero_page);

    const T* a_row[4];
    int cb[4], cy[4], cx[4];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + lr + j * 32;
            m = m < p.M ? m : p.M - 1;
            a_row[j] = reinterpret_cast<const T*>(p.A) + (size_t)m * p.lda;
        }
    } else {
        const int hw = p.g.Ho * p.g.Wo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + lr + j * 32;
            m = m < p.M ? m : p.M - 1;
            const int b = m / hw, r = m - b * hw;
            const int y = r / p.g.Wo;
            cb[j] = b;
            cy[j] = y * p.g.stride - 1;
            cx[j] = (r - y * p.g.Wo) * p.g.stride - 1;
        }
    }
    // this wave's W rows: passes grp, grp+2, ...
    const T* w_row[WP0];
#pragma unroll
    for (int j = 0; j < WP0; ++j) {
        int n = n0 + lr + (2 * j + grp) * 32;
        n = n < p.N ? n : p.N - 1;
        w_row[j] = Wp + (size_t)n * p.ldw;
    }

    const int Ctot = p.g.C1 + p.g.C2;
    int tap = 0, cin = lc * 8;
    if constexpr (MODE == 1) {
        if (!p.g.chunk64)
            while (cin >= Ctot) { cin -= Ctot; ++tap; }
    }

    auto dma_tile = [&](int kt) {       // tiles must be requested in increasing kt order (tap-major conv state)
        const bool kin = kt * BK + lc * 8 < p.K;
        unsigned char* st = smem + (kt % NSTAGE) * STAGE;
        unsigned char* As = st + grp * A_STAGE;
        unsigned char* Ws = st + 2 * A_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T* s;
            if constexpr (MODE == 0) {
                s = kin ? a_row[j] + kt * BK + lc * 8 : zero;
            } else {
                int t_ = tap, c_ = cin;
                if (p.g.chunk64) { t_ = kt % 9; c_ = (kt / 9) * 64 + lc * 8; }
                s = conv_src<T>(p, cb[j], cy[j], cx[j], t_, c_, kin);
                s = s ? s : zero;
            }
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)s, (lds_ptr_t)(As + (j * 32 + gw * 8) * ROW_BYTES), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WP0; ++j) {
            if (j < (grp == 0 ? WP0 : WP1)) {
                const T* s = kin ? w_row[j] + kt * BK + lc * 8 : zero;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)s, (lds_ptr_t)(Ws + ((2 * j + grp) * 32 + gw * 8) * ROW_BYTES), 16, 0, 0);
            }
        }
        if constexpr (MODE == 1) {
            if (!p.g.chunk64) {
                cin += BK;
                while (cin >= Ctot) { cin -= Ctot; ++tap; }
            }
        }
    };

    f32x4 acc[NF][MF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    const int frow = lane & 15, fchunk = lane >> 4;
    // lane-constant fragment offsets: the swizzle term of row (base + 16 i + frow) equals that of frow
    int xoff[2], woff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        xoff[ks] = swz(wm * 64 + frow, ks * 4 + fchunk);
        woff[ks] = swz(wn * WN + frow, ks * 4 + fchunk);
    }

    // ---- prologue: tiles 0 and 1 in flight, landed and visible before anybody reads ---------------------------
    dma_tile(0);
    if (nk > 1) dma_tile(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 1) {                     // stagger: group 1 now runs one phase behind group 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    for (int kt = 0; kt < nk; ++kt) {
        // ================= phase L: DMA for kt+2, fragments of kt -> registers ===============================
        const bool more = kt + 2 < nk;
        if (more) dma_tile(kt + 2);
        const unsigned char* st = smem + (kt % NSTAGE) * STAGE;
        const unsigned char* As = st + grp * A_STAGE;
        const unsigned char* Ws = st + 2 * A_STAGE;
        V8 xf[2][MF], wf[2][NF];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < MF; ++i) xf[ks][i] = *reinterpret_cast<const V8*>(As + xoff[ks] + i * 16 * ROW_BYTES);
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[ks][j] = *reinterpret_cast<const V8*>(Ws + woff[ks] + j * 16 * ROW_BYTES);
        }
        // this wave's DMA of tile kt+1 (issued one iteration ago) must have landed; the DMA just issued may stay in flight
        if (more) {
            if (grp == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA0) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ================= phase M: MFMAs on registers only ====================================================
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int i = 0; i < MF; ++i) acc[j][i] = Tag::mfma16(wf[ks][j], xf[ks][i], acc[j][i]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (grp == 0) {                     // re-align the groups: group 0 waits for group 1's last phase M
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __syncthreads();

    // ---- epilogue: per group, two 64-row passes through an fp32 LDS tile -------------------------------------------
    float* Cs = reinterpret_cast<float*>(smem + grp * CS_BYTES);
    constexpr int CHUNKS = BN / 8;
    constexpr int TASKS = 64 * CHUNKS;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const int r = i * 16 + (lane & 15);
                    const int c = wn * WN + j * 16 + (lane >> 4) * 4;
                    *reinterpret_cast<f32x4*>(Cs + r * CS_LD + c) = acc[j][i];
                }
        }
        __syncthreads();
        for (int task = gt; task < TASKS; task += 256) {
            const int r = task / CHUNKS, ch = task - r * CHUNKS;
            const int m = m0 + pass * 64 + r, n = n0 + ch * 8;
            if (m >= p.M || n >= p.N) continue;
            float v[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + r * CS_LD + ch * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + r * CS_LD + ch * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
            gemm_epilogue_store<Tag>(p, m, n, v);
        }
        __syncthreads();
    }
}

template <class Tag, int MODE>
int launch_rs(const GemmParams& p, hipStream_t s) {
    int bn = 128;
    if (p.N % 160 == 0) bn = 160;
    else if (p.N % 128 == 0) bn = 128;
    else if (p.N <= 64) bn = 64;
    const unsigned grid = mve_cdiv(p.M, 256) * mve_cdiv(p.N, bn);
    if (bn == 160) k_gemm_rs<Tag, 160, MODE><<<grid, NT2, 0, s>>>(p);
    else if (bn == 128) k_gemm_rs<Tag, 128, MODE><<<grid, NT2, 0, s>>>(p);
    else k_gemm_rs<Tag, 64, MODE><<<grid, NT2, 0, s>>>(p);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

}  // namespace

int mve_gemm_rs_launch(int dtype, int mode, const void* params, void* stream) {
    const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MVE_F16) return mode == 0 ? launch_rs<F16Tag, 0>(p, s) : launch_rs<F16Tag, 1>(p, s);
    if (dtype == MVE_BF16) return mode == 0 ? launch_rs<BF16Tag, 0>(p, s) : launch_rs<BF16Tag, 1>(p, s);
    mve_set_error("gemm_rs: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}
