// 256 x 320 x 64 tile variant of the MFMA GEMM / implicit-GEMM conv (see gemm.hip for the algorithm, the LDS layout, the
// swapped-operand MFMA and the epilogue conventions; this file only changes the decomposition).
//
// Why: rocprofv3 PMC passes on the 128 x 160 kernel (profiles/r01_rocprof_v3_summary.txt) show conv launches fetching ~5x
// their algorithmic bytes (operand re-reads across N- and M-tiles miss L2), and per k-tile the four 64 x 80 wave tiles read
// 72 KB of fragments from LDS for 640 MFMA cycles per wave -- LDS bandwidth (128 B/clk/CU) is ~90 % of the MFMA time.
//   block 256 x 320, 512 threads = 8 waves as 2 (M) x 4 (N); wave tile 128 x 80 = 8 x 5 fragments, 160 accumulator registers;
//   LDS fragment traffic per MFMA drops by 1.4x ((128+80)/(128*80) vs (64+80)/(64*80)), global->LDS traffic per FLOP by 2x;
//   two 72 KB stages = 144 KB of the 160 KB LDS (dynamic shared memory), one block per CU, 2 waves per SIMD.
// The k order (tile by tile, two 32-wide MFMA steps per tile) and the split-K slicing are identical to gemm.hip, so both
// kernels produce bit-identical results and the dispatcher may pick either by problem size without breaking batch invariance.
#include "common.h"

#include "gemm_shared.h"
#include "gemm_big_epilogue.h"

namespace {

// The tile width BN2 is a template parameter: 320 (wave tile 128 x 80, every UNet width is a multiple of 320) or 256 (wave tile
// 128 x 64: the VAE's 256 / 512-channel convolutions, whose N is not a multiple of 320).
constexpr int BM2 = 256, NTH = 512;
constexpr int WAVES_N = 4;
constexpr int WTM = 128;
constexpr int MF = WTM / 16;                         // 8 fragments per wave along M
constexpr int RPP = NTH / 8;                         // 64 rows per load pass
constexpr int A_PASSES = BM2 / RPP;                  // 4
constexpr int A_STAGE = BM2 * ROW_BYTES;
constexpr int smem_big(int bn2) { return 2 * (A_STAGE + bn2 * ROW_BYTES); }      // 320: 147456 B; the fp32 epilogue tile lives inside it
static_assert(64 * (320 + 4) * 4 <= smem_big(320) && 64 * (256 + 4) * 4 <= smem_big(256), "epilogue staging must fit");

template <class Tag, int MODE, bool SEQ, bool FAST, int BN2>      // FAST (conv only): slab-major K order
__global__ __launch_bounds__(NTH, 2) void k_gemm_big(const GemmParams p) {
    typedef typename Tag::V8 V8;
    typedef typename Tag::T T;
    constexpr int WTN = BN2 / WAVES_N, NF = WTN / 16;          // 80 -> 5 fragments, 64 -> 4
    constexpr int B_PASSES = BN2 / RPP;                        // 5 | 4
    constexpr int STAGE = A_STAGE + BN2 * ROW_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;

    const int tiles_n = (p.N + BN2 - 1) / BN2;
    const int tiles_m = (p.M + BM2 - 1) / BM2;
    const int S = p.splitk > 1 ? p.splitk : 1;
    const unsigned lin = mve_xcd_remap(blockIdx.x, (unsigned)(tiles_m * tiles_n * S));
    const int kslice = lin % S;
    const unsigned tile = lin / S;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;

    const int lr = tid >> 3;                                   // 0..63
    const int lc = (tid & 7) ^ ((lr >> 1) & 7);                // source chunk: swizzle applied on the global side
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.W);

    const T* a_row[A_PASSES];
    int cb[A_PASSES], cy[A_PASSES], cx[A_PASSES];
    int pix[A_PASSES];
    unsigned vmask[A_PASSES];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            int m = m0 + lr + j * RPP;
            m = m < p.M ? m : p.M - 1;
            a_row[j] = reinterpret_cast<const T*>(p.A) + (size_t)m * p.lda;
        }
    } else {
        const int hw = p.g.Ho * p.g.Wo;
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            int m = m0 + lr + j * RPP;
            m = m < p.M ? m : p.M - 1;
            const int b = m / hw, r = m - b * hw;
            const int y = r / p.g.Wo;
            cb[j] = b;
            cy[j] = y * p.g.stride - p.g.pad;
            cx[j] = (r - y * p.g.Wo) * p.g.stride - p.g.pad_x;
            if constexpr (FAST) {     // per-row pixel base and 9-bit halo mask; cb/cy/cx are dead after this in the fast kernel
                // nearest 2x upsample: the source pixel of virtual (y, x) is (y >> 1, x >> 1); bits 9 / 10 keep the parities of the
                // window origin, from which a tap's source offset is ((parity + d) >> 1)
                pix[j] = (cb[j] * p.g.Hs + (cy[j] >> p.g.ups)) * p.g.Ws + (cx[j] >> p.g.ups);
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
    }
    const T* w_row[B_PASSES];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
        int n = n0 + lr + j * RPP;
        n = n < p.N ? n : p.N - 1;
        w_row[j] = Wp + (size_t)n * p.ldw;
    }

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

    // Fast conv addressing (slab-major K order, no upsample): the im2col source of (row j, tap t, slab) is
    //   src + ((pix[j] + dy*Ws + dx) * cs + ch) with everything but pix[j] uniform per K tile, and the halo test is one bit of
    // a per-row 9-bit mask computed once.  The generic path below costs ~45 VALU instructions and two divergent branches per
    // row and K tile (64-bit multiplies, bounds tests); with two waves per SIMD running the loop in lock-step that address
    // phase left the MFMA pipe idle for a third of every K tile (rocprofv3 SQ_WAIT_* / ISA inspection, round 1).
    auto dma_tile = [&](int kt, int stage) {
        const bool kin = kt * BK + lc * 8 < p.K;
        unsigned char* As = smem + stage * STAGE;
        unsigned char* Bs = As + A_STAGE;
        if constexpr (MODE == 1 && FAST) {
            int t_ = kt % 9, c0 = (kt / 9) * 64;                       // uniform
            bool second = c0 >= p.g.C1;
            const T* src = reinterpret_cast<const T*>(second ? p.A2 : p.A);
            int cs = second ? p.g.C2 : p.g.C1;
            int ch = (second ? c0 - p.g.C1 : c0) + lc * 8;
            if (p.g.nk_main > 0 && kt >= p.g.nk_main) {                // 1x1 shortcut part: centre tap of the shortcut sources
                t_ = 4;
                c0 = (kt - p.g.nk_main) * 64;
                second = c0 >= p.g.C3;
                src = reinterpret_cast<const T*>(second ? p.A4 : p.A3);
                cs = second ? p.g.C4 : p.g.C3;
                ch = (second ? c0 - p.g.C3 : c0) + lc * 8;
            }
            const int dy = t_ / 3, dx = t_ - dy * 3;
            const int toff = dy * p.g.Ws + dx;
#pragma unroll
            for (int j = 0; j < A_PASSES; ++j) {
                int to = toff;
                if (p.g.ups) to = (int)((((vmask[j] >> 9) & 1u) + dy) >> 1) * p.g.Ws + (int)((((vmask[j] >> 10) & 1u) + dx) >> 1);
                const unsigned off = (unsigned)((pix[j] + to) * cs + ch);
                const T* s = ((vmask[j] >> t_) & 1u) ? src + off : zero;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)s, (lds_ptr_t)(As + (j * RPP + wid * 8) * ROW_BYTES), 16, 0, 0);
            }
        } else {
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            const T* s;
            if constexpr (MODE == 0) s = kin ? a_row[j] + kt * BK + lc * 8 : nullptr;
            else {
                int t_ = tap, c_ = cin;
                if (p.g.chunk64) { t_ = kt % 9; c_ = (kt / 9) * 64 + lc * 8; }
                s = conv_src<T>(p, cb[j], cy[j], cx[j], t_, c_, kin);
            }
            s = s ? s : zero;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)s, (lds_ptr_t)(As + (j * RPP + wid * 8) * ROW_BYTES), 16, 0, 0);
        }
        }
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) {
            const T* s = kin ? w_row[j] + kt * BK + lc * 8 : zero;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)s, (lds_ptr_t)(Bs + (j * RPP + wid * 8) * ROW_BYTES), 16, 0, 0);
        }
        if constexpr (MODE == 1 && !FAST) {
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

    dma_tile(kt_begin, 0);
    __syncthreads();

    // sequential split-K emulation (see GemmParams::splitk_seq): the K loop runs slice by slice; between slices the accumulators
    // are folded into lane-private 16-byte slots of a block-private fp32 running total (1 KiB per wave access)
    const int SQ = SEQ ? p.splitk_seq : 1;
    const int frow = lane & 15, fchunk = lane >> 4;
    int kt = kt_begin;
    for (int sl = 0; sl < SQ; ++sl) {
        const int kend = SEQ ? (int)((long long)nk_all * (sl + 1) / SQ) : kt_end;
        for (; kt < kend; ++kt) {
            const int cur = (kt - kt_begin) & 1;
            const unsigned char* As = smem + cur * STAGE;
            const unsigned char* Bs = As + A_STAGE;
            auto mfma_step = [&](int ks) {
                V8 xf[MF];
#pragma unroll
                for (int i = 0; i < MF; ++i) xf[i] = *reinterpret_cast<const V8*>(As + swz(wm * WTM + i * 16 + frow, ks * 4 + fchunk));
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const V8 wf = *reinterpret_cast<const V8*>(Bs + swz(wn * WTN + j * 16 + frow, ks * 4 + fchunk));
#pragma unroll
                    for (int i = 0; i < MF; ++i) acc[j][i] = Tag::mfma16(wf, xf[i], acc[j][i]);
                }
            };
            // (Tried: staggering the DMA issue of the two waves that share a SIMD -- "ping-pong" -- measured 1 % slower end to end
            // on the same box, profiles/r01_ab_tuning.log, so both wave groups issue at the top of the iteration.)
            if (kt + 1 < kt_end) dma_tile(kt + 1, cur ^ 1);
            mfma_step(0);
            mfma_step(1);
            __syncthreads();   // drains the in-flight LDS-DMA of tile kt+1 (vmcnt(0)) and frees stage `cur`
        }
        if constexpr (SEQ) {
            f32x4* tot = reinterpret_cast<f32x4*>(p.partial) + ((size_t)tile * (BM2 * BN2 / 4) + (size_t)wid * (NF * MF * 64) + lane);
            const bool last = sl + 1 == SQ;
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    f32x4* slot = tot + (j * MF + i) * 64;
                    f32x4 t = acc[j][i];
                    if (sl > 0) { const f32x4 o = *slot; t = f32x4{o[0] + t[0], o[1] + t[1], o[2] + t[2], o[3] + t[3]}; }
                    if (last) acc[j][i] = t;                       // result = running total + last slice
                    else { *slot = t; acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }

    big_tile_epilogue<Tag, BN2>(p, acc, smem, m0, n0, kslice, tid, lane, wm, wn);
}

int big_bn(int N) { return N % 320 == 0 ? 320 : (N % 256 == 0 ? 256 : 0); }

template <class Tag, int MODE, bool SEQ, bool FAST, int BN2>
int launch_big3(const GemmParams& p, hipStream_t s) {
    static bool configured[64] = {};             // the attribute is per device; one process may drive several
    int dev = 0;
    MVE_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !configured[dev]) {
        MVE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_big<Tag, MODE, SEQ, FAST, BN2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    smem_big(BN2)));
        configured[dev] = true;
    }
    const unsigned grid = (unsigned)mve_cdiv(p.M, BM2) * (unsigned)mve_cdiv(p.N, BN2) * (unsigned)(p.splitk > 1 ? p.splitk : 1);
    k_gemm_big<Tag, MODE, SEQ, FAST, BN2><<<grid, NTH, smem_big(BN2), s>>>(p);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}
template <class Tag, int MODE, bool SEQ, int BN2>
int launch_big2(const GemmParams& p, hipStream_t s) {
    if constexpr (MODE == 1) {
        if (p.g.chunk64) return launch_big3<Tag, MODE, SEQ, true, BN2>(p, s);
    }
    return launch_big3<Tag, MODE, SEQ, false, BN2>(p, s);
}
template <class Tag, int MODE>
int launch_big(const GemmParams& p, hipStream_t s) {
    if (big_bn(p.N) == 256) {      // no split-K variants of the 256-wide tile (mve_gemm_big_blocks reports such shapes as not eligible)
        MVE_CHECK(p.splitk <= 1 && p.splitk_seq <= 1, MVE_ERR_ARG, "gemm_big: the 256-wide tile does not split K");
        return launch_big2<Tag, MODE, false, 256>(p, s);
    }
    return p.splitk_seq > 1 ? launch_big2<Tag, MODE, true, 320>(p, s) : launch_big2<Tag, MODE, false, 320>(p, s);
}

}  // namespace

// number of blocks the 256 x 320 kernel would launch (0: shape not eligible)
long long mve_gemm_big_blocks(int M, int N, int splitk) {
    const int bn = big_bn(N);
    if (bn == 0 || M < 64 || (bn == 256 && splitk > 1)) return 0;
    return (long long)mve_cdiv(M, BM2) * (N / bn) * (splitk > 1 ? splitk : 1);
}

// main loop + epilogue (or split-K partials; the caller runs the reducer)
int mve_gemm_big_launch(int dtype, int mode, const void* params, void* stream) {
    const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MVE_F16) return mode == 0 ? launch_big<F16Tag, 0>(p, s) : launch_big<F16Tag, 1>(p, s);
    if (dtype == MVE_BF16) return mode == 0 ? launch_big<BF16Tag, 0>(p, s) : launch_big<BF16Tag, 1>(p, s);
    mve_set_error("gemm_big: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}
