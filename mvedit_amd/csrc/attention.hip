// Flash-style scaled-dot-product attention for gfx950 (fp16 / bf16 storage, fp32 softmax + accumulate).
//
//   O[b,i,h,:] = softmax_j( scale * Q[b,i,h,:] . K[b,j,h,:] ) V[b,j,h,:]
//
// replaces F.scaled_dot_product_attention as called by the reference's vendored processors
// (lib/models/architecture/ip_adapter/attention_processor.py:246-248, :348-350, :364-366) for
//   - self attention (L = HW, or 2*HW under CrossImageAttnProcWrapper, joint_attn.py:13-17: a pure
//     re-interpretation of [2b, L, C] as [b, 2L, C], i.e. just B and L here);
//   - cross attention to text (Lk = 77) and the IP-Adapter branch (Lk = 16);
//   - reference attention (zero123plus.py:66-69, diffusers.py:646-673): keys/values are the concatenation
//     of two sources, passed as two (pointer, length) segments so nothing is materialised.
// Q/K/V/O are strided views into the packed projection outputs ([B*L, ld] rows, head h at column h*d),
// so no head split / merge permutes exist anywhere.
//
// Kernel shape (wave64, v_mfma_f32_16x16x32):
//   block = 4 waves x 32 query rows; KV tile = 128 keys (64 for d > 64), staged once per block in LDS.
//   Both products are issued TRANSPOSED so that no cross-lane transpose of P is ever needed:
//       S^T = K Q^T   (A operand = K fragment from LDS, B operand = Q fragment held in registers)
//       O^T = V^T P^T (A operand = V^T fragment from LDS, B operand = P^T built in-lane from S^T)
//   In S^T the lane's column is its query (lane&15) and its 4 accumulator rows are 4 consecutive keys, so
//   a lane already owns the 8 P values the second MFMA wants for its query -- provided the contraction
//   slots of that MFMA are *defined* as "keys 4g..4g+3 of key-fragment 2kk, then of fragment 2kk+1".  MFMA
//   sums over slots, so any slot<->key assignment is legal as long as both operands use the same one; V^T is
//   therefore written to LDS with its keys permuted to match (done for free while transposing).
//   Row statistics (max / sum over keys) are per-lane over registers plus two xor-shuffles (lanes l, l^16,
//   l^32, l^48 share a query); the output rescale factor is lane-local.
//   V is transposed while it is staged (two keys per thread -> packed 32-bit LDS writes).
//   LDS images use 16-byte-chunk XOR swizzles so the ds_read_b128 fragment reads are (near) conflict free.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int NT = 256;      // 4 waves; a wave owns QF fragments of 16 query rows (QF = 2: 128 query rows per block)

struct AttnParams {
    const void* Q; const void* K; const void* V; const void* K2; const void* V2; void* O;
    int ldq, ldk, ldv, ldk2, ldv2, ldo;
    int B, Lq, Lk, Lk2, heads;
    float scale_log2e;   // softmax scale * log2(e); 1 when the caller folded it into Q (mve_attention_prescaled)
    int prescaled;
};

// chunk permutation of K row `row` (a bijection inside every group of 4 chunks, applied identically by the LDS-DMA source side and the
// fragment reads).  DP = 64: 128-byte rows, conflict-free.  Other DP (192- / 320-byte rows): the original term (row >> 2) & 3 leaves the
// ds_read_b128 lane groups {0-3,12-15,20-27}, ... 2-way conflicted; KSW2 uses ((row >> 3) & 1) << 1, which an exhaustive search over the
// per-4-row tables shows to be conflict-free for both strides (an experiment behind mve_attention_tune(2) until it has run on hardware).
template <int DP, bool KSW2> __device__ __forceinline__ int k_perm(int row, int pc) {
    if constexpr (DP == 64) return pc ^ ((row >> 1) & 7);
    else if constexpr (KSW2) return pc ^ (((row >> 3) & 1) << 1);
    else return pc ^ ((row >> 2) & 3);
}
template <int DP, bool KSW2> __device__ __forceinline__ int k_swz(int row, int chunk) {
    // K tile: [64 keys][DP] 16-bit, DP*2 bytes per row
    return row * (DP * 2) + (k_perm<DP, KSW2>(row, chunk) << 4);
}

// ---------------------------------------------------------------------------------------------------------
// Pipeline (the math and fragment conventions are the ones described in the header; a first version with 64-key fills,
// register-staged K and an unconditional rescale measured 339 TFLOP/s at d = 40 against 472 for this one):
//   * LDS is filled 128 keys at a time (64 for d = 160) and consumed as 64-key halves: ONE barrier per fill;
//   * K tiles arrive by LDS-DMA (global_load_lds_dwordx4, swizzle applied to the source chunk, pad chunks and
//     out-of-range keys sourced from a zero page) into a double buffer, issued a whole tile ahead;
//   * V^T is double buffered too: its global loads are issued at the top of the tile, transposed into the
//     other buffer after the MFMAs, so nothing waits on HBM inside a tile;
//   * the running-max rescale of O is skipped (wave-uniform branch) whenever no query's maximum grew -- exact,
//     not a thresholded approximation; after the first few tiles that is almost always;
//   * P is rounded to the storage dtype with packed converts (v_cvt_pk_{f16,bf16}_f32);
//   * VALU diet (rocprofv3: the first version issued 8.7 VALU per MFMA and was VALU-issue bound): every LDS address
//     and DMA slot is a lane constant computed once; the key mask lives in a separately compiled half; for d = 40
//     a row of ones in the spare rows of the last V^T fragment makes the PV MFMA accumulate the softmax denominator.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) unsigned int g_attn_zero_page[4] = {0u, 0u, 0u, 0u};

// V^T tile: [dv rows][KB2 keys] 16-bit.  The fragment reads (ds_read_b128) are conflict-free with the base term alone, but the transposing
// ds_write_b32 stores of store_v are not: the 32 lanes of a store group hold 5-8 different dv chunks c = row >> 3 whose base terms
// coincide, so they pile onto the same banks (tools/lds_conflicts.py: 5- to 10-way).  VSW2 adds a term in c that keeps the reads
// conflict-free and cuts the stores to 2- to 4-way (an experiment behind mve_attention_tune(4) until it has run on hardware).
template <int KB2, bool VSW2> __device__ __forceinline__ int v_swz2(int row, int chunk) {
    if constexpr (KB2 == 128) return row * 256 + ((chunk ^ (row & 15) ^ (VSW2 ? (row >> 3) & 7 : 0)) << 4);   // 256-byte rows
    else return row * 128 + ((chunk ^ ((row >> 1) & 7) ^ (VSW2 ? (row >> 4) & 7 : 0)) << 4);
}

// QF, KB2X, OCC: experiment knobs (mve_attention_tune).  The defaults (2, 0, 0) are the measured configuration; QF = 1 halves the
// per-wave state (16 query rows), KB2X overrides the keys per LDS fill, OCC the blocks per CU the register allocator targets.
template <class Tag, int D, bool SEG2, int QF = 2, int KB2X = 0, int OCC = 0, bool KSW2 = false, bool VSW2 = false>
__global__ __launch_bounds__(NT, (OCC ? OCC : (D > 80 ? 1 : 2))) void k_attention2(const AttnParams p) {
    constexpr int QB = 64 * QF;      // query rows per block
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int KS = DP / 32;
    constexpr int DC = D / 8;
    constexpr int DVF = (D + 15) / 16;
    constexpr int KB2 = KB2X ? KB2X : (D > 64 ? 64 : 128);       // keys per LDS fill (LDS budget: 2 blocks/CU for d = 80)
    constexpr int NH = KB2 / 64;                 // 64-key halves per fill
    constexpr int CPR = DP / 8;                  // 16-byte chunks per K row
    constexpr int KROW = DP * 2;                 // bytes per K row
    constexpr int VROW = KB2 * 2;                // bytes per V^T row
    constexpr int K_BYTES = KB2 * KROW;
    constexpr int V_BYTES = DVF * 16 * VROW;
    constexpr int K_INSTR = K_BYTES / 1024 / 4;  // LDS-DMA instructions per wave per fill (1 KiB each)
    static_assert((K_BYTES / 1024) % 4 == 0, "K tile must split evenly over 4 waves");
    constexpr int V_TASKS = (KB2 / 2) * DC;
    constexpr int V_PER_T = (V_TASKS + NT - 1) / NT;
    // When D is not a multiple of 16 the last V^T fragment has spare rows: row D is set to all ones, so the PV MFMA
    // itself produces l = sum_j P (rounded exactly like the numerator) and the 32 adds per half disappear.
    constexpr bool ONES = (D % 16) != 0;
    constexpr int L_G = (D % 16) / 4, L_R = (D % 16) % 4;
    typedef typename Tag::V8 V8;
    typedef typename Tag::T T;

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * K_BYTES + 2 * V_BYTES];
    unsigned char* Kbuf = smem;
    unsigned char* Vbuf = smem + 2 * K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;

    const int q_tiles = (p.Lq + QB - 1) / QB;
    const unsigned nblk = (unsigned)(q_tiles * p.heads * p.B);
    const unsigned bid = mve_xcd_remap(blockIdx.x, nblk);
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % p.heads;
    const int b = bid / (q_tiles * p.heads);
    const int Ltot = p.Lk + p.Lk2;

    const T* Qp = reinterpret_cast<const T*>(p.Q);
    const T* zero = reinterpret_cast<const T*>(g_attn_zero_page);
    // per-(batch, head) bases; key j of segment 1 lives at kb1 + j*ldk
    const T* kb1 = reinterpret_cast<const T*>(p.K) + (size_t)b * p.Lk * p.ldk + h * D;
    const T* vb1 = reinterpret_cast<const T*>(p.V) + (size_t)b * p.Lk * p.ldv + h * D;
    const T* kb2 = SEG2 ? reinterpret_cast<const T*>(p.K2) + (size_t)b * p.Lk2 * p.ldk2 + h * D : nullptr;
    const T* vb2 = SEG2 ? reinterpret_cast<const T*>(p.V2) + (size_t)b * p.Lk2 * p.ldv2 + h * D : nullptr;
    auto k_row = [&](int j) -> const T* {       // j already clamped to [0, Ltot)
        if constexpr (SEG2) { if (j >= p.Lk) return kb2 + (size_t)(j - p.Lk) * p.ldk2; }
        return kb1 + (size_t)j * p.ldk;
    };
    auto v_row = [&](int j) -> const T* {
        if constexpr (SEG2) { if (j >= p.Lk) return vb2 + (size_t)(j - p.Lk) * p.ldv2; }
        return vb1 + (size_t)j * p.ldv;
    };

    // V^T buffers: zero (rows dv >= D contribute nothing), then the all-ones row that accumulates l
    for (int i = tid; i < 2 * V_BYTES / 16; i += NT) reinterpret_cast<u32x4*>(Vbuf)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if constexpr (ONES) {
        const unsigned short one = __builtin_bit_cast(unsigned short, Tag::from_f32(1.0f));
        const unsigned two = (unsigned)one | ((unsigned)one << 16);
        for (int i = tid; i < 2 * (VROW / 4); i += NT) {
            const int buf = i / (VROW / 4), wofs = i - buf * (VROW / 4);
            *reinterpret_cast<unsigned*>(Vbuf + buf * V_BYTES + D * VROW + wofs * 4) = two;   // whole row: swizzle only permutes it
        }
    }

    V8 qf[QF][KS];
    const int q_base = qt * QB + wid * (16 * QF);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        int q = q_base + f * 16 + l16;
        q = q < p.Lq ? q : p.Lq - 1;
        const T* row = Qp + ((size_t)b * p.Lq + q) * p.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 4 + g;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c < DC) v = *reinterpret_cast<const u32x4*>(row + c * 8);
            qf[f][ks] = __builtin_bit_cast(V8, v);
        }
    }

    // ---- lane constants: LDS-DMA slots of the K tile, V^T staging tasks, fragment read offsets -----------------
    int kd_row[K_INSTR], kd_col[K_INSTR];        // key row inside the tile, element offset of the logical chunk (-1: pad)
#pragma unroll
    for (int i = 0; i < K_INSTR; ++i) {
        const int pos = (wid * K_INSTR + i) * 64 + lane;
        const int row = pos / CPR, pc = pos - row * CPR;
        const int c = k_perm<DP, KSW2>(row, pc);
        kd_row[i] = row;
        kd_col[i] = c < DC ? c * 8 : -1;
    }
    int vt_key[V_PER_T], vt_col[V_PER_T], vt_lds[V_PER_T];
#pragma unroll
    for (int i = 0; i < V_PER_T; ++i) {
        const int task = tid + i * NT;
        const int pair = task / DC, c = task - pair * DC;
        const int key = 2 * pair;                 // key = 32*kk + 16*hh + 4*gg + jj  ->  slot 32*kk + 8*gg + 4*hh + jj
        const int kk = key >> 5, hh = (key >> 4) & 1, gg = (key >> 2) & 3, jj = key & 3;
        const int pos = 32 * kk + 8 * gg + 4 * hh + jj;
        vt_key[i] = task < V_TASKS ? key : -1;
        vt_col[i] = c * 8;
        vt_lds[i] = ((pos >> 3) << 16) | ((pos & 7) * 2);   // (chunk, byte offset inside the chunk)
    }
    // K fragment (A operand) of key fragment kf, k-step ks, half hf:  koff[ks] + (hf*64 + kf*16) * KROW
    // the swizzle term depends on (row>>1)&7 resp. (row>>2)&3, which equals the lane's own l16 term because the
    // fragment row offsets are multiples of 16.
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = k_swz<DP, KSW2>(l16, ks * 4 + g);
    // V^T fragment (A operand) of dv fragment i, key step (hf, kk):  voff[hf*2+kk] + i*16*VROW.  With VSW2 the swizzle term depends on
    // the fragment too: row = 16 i + l16 changes it by the constant XOR VX(i), kept as per-(i, step) lane constants when that is cheap
    // in registers (VTAB) and applied at the read otherwise.
    constexpr bool VTAB = VSW2 && DVF * NH * 2 <= 12;
    int voff[VTAB ? 1 : NH * 2], voff2[VTAB ? DVF : 1][NH * 2];
    if constexpr (VTAB) {
#pragma unroll
        for (int i = 0; i < DVF; ++i)
#pragma unroll
            for (int s2 = 0; s2 < NH * 2; ++s2) voff2[i][s2] = v_swz2<KB2, true>(i * 16 + l16, s2 * 4 + g);
    } else {
#pragma unroll
        for (int s2 = 0; s2 < NH * 2; ++s2) voff[s2] = v_swz2<KB2, VSW2>(l16, s2 * 4 + g);
    }
    auto vx = [](int i) { return !VSW2 ? 0 : (KB2 == 128 ? ((2 * i) & 7) << 4 : (i & 7) << 4); };

    auto dma_k = [&](int t, int buf) {
        const int j0 = t * KB2;
#pragma unroll
        for (int i = 0; i < K_INSTR; ++i) {
            int j = j0 + kd_row[i];
            j = j < Ltot ? j : Ltot - 1;                       // clamped rows hold finite data; their scores are masked
            const T* src = kd_col[i] >= 0 ? k_row(j) + kd_col[i] : zero;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(Kbuf + buf * K_BYTES + (wid * K_INSTR + i) * 1024), 16, 0, 0);
        }
    };
    u32x4 vreg[V_PER_T][2];
    auto load_v = [&](int t) {
        const int j0 = t * KB2;
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) {
            if (vt_key[i] >= 0) {
                int j = j0 + vt_key[i];
                const int ja = j < Ltot ? j : Ltot - 1, jb = j + 1 < Ltot ? j + 1 : Ltot - 1;
                vreg[i][0] = *reinterpret_cast<const u32x4*>(v_row(ja) + vt_col[i]);
                vreg[i][1] = *reinterpret_cast<const u32x4*>(v_row(jb) + vt_col[i]);
            }
        }
    };
    auto store_v = [&](int buf) {
        unsigned char* Vs = Vbuf + buf * V_BYTES;
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) {
            if (vt_key[i] >= 0) {
                const int chunk = vt_lds[i] >> 16, within = vt_lds[i] & 0xffff;
                const unsigned short* a = reinterpret_cast<const unsigned short*>(&vreg[i][0]);
                const unsigned short* bb = reinterpret_cast<const unsigned short*>(&vreg[i][1]);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *reinterpret_cast<unsigned*>(Vs + v_swz2<KB2, VSW2>(vt_col[i] + e, chunk) + within) = (unsigned)a[e] | ((unsigned)bb[e] << 16);
            }
        }
    };

    f32x4 oacc[DVF][QF];
#pragma unroll
    for (int i = 0; i < DVF; ++i)
#pragma unroll
        for (int f = 0; f < QF; ++f) oacc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) { m_run[f] = -INFINITY; l_run[f] = 0.f; }

    // one 64-key half: S^T = K Q^T, online softmax, O^T += V^T P^T.  MASKED is compiled separately so that the common
    // path carries no select instructions (the compiler if-converts a runtime test into 48 v_cndmask per half).
    auto do_half = [&](const unsigned char* Ks, const unsigned char* Vs, int hf, int key0, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        f32x4 s[4][QF];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int f = 0; f < QF; ++f) s[kf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const V8 ka = *reinterpret_cast<const V8*>(Ks + koff[ks] + (hf * 64 + kf * 16) * KROW);
#pragma unroll
                for (int f = 0; f < QF; ++f) s[kf][f] = Tag::mfma16(ka, qf[f][ks], s[kf][f]);
            }
        }
        if constexpr (MASKED) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key0 + kf * 16 + g * 4 + r >= Ltot) {
#pragma unroll
                        for (int f = 0; f < QF; ++f) s[kf][f][r] = -INFINITY;
                    }
        }
        V8 pf[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float mx = -INFINITY;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kf][f][r]);
            mx = mve_max_xor32(mve_max_xor16(mx));
            const float mxs = mx * p.scale_log2e;
            if (__any(mxs > m_run[f])) {                  // some query's running maximum grows: rescale (exact)
                const float m_new = fmaxf(m_run[f], mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
                m_run[f] = m_new;
                if constexpr (!ONES) l_run[f] *= alpha;
#pragma unroll
                for (int i = 0; i < DVF; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[i][f][r] *= alpha;
            }
            const float mr = m_run[f];
            float psum = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x8 e;
                {
                    // exponent arguments on the packed-fp32 pipe (v_pk_fma_f32: two lanes of work per instruction)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 sc2 = {p.scale_log2e, p.scale_log2e}, mr2 = {mr, mr};
                    const f32x4 a4 = s[2 * kk][f], b4 = s[2 * kk + 1][f];
                    const f32x2 t0 = __builtin_elementwise_fma(f32x2{a4[0], a4[1]}, sc2, -mr2);
                    const f32x2 t1 = __builtin_elementwise_fma(f32x2{a4[2], a4[3]}, sc2, -mr2);
                    const f32x2 t2 = __builtin_elementwise_fma(f32x2{b4[0], b4[1]}, sc2, -mr2);
                    const f32x2 t3 = __builtin_elementwise_fma(f32x2{b4[2], b4[3]}, sc2, -mr2);
                    e[0] = __builtin_amdgcn_exp2f(t0[0]); e[1] = __builtin_amdgcn_exp2f(t0[1]);
                    e[2] = __builtin_amdgcn_exp2f(t1[0]); e[3] = __builtin_amdgcn_exp2f(t1[1]);
                    e[4] = __builtin_amdgcn_exp2f(t2[0]); e[5] = __builtin_amdgcn_exp2f(t2[1]);
                    e[6] = __builtin_amdgcn_exp2f(t3[0]); e[7] = __builtin_amdgcn_exp2f(t3[1]);
                }
                if constexpr (!ONES) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) psum += e[r];
                }
                pf[f][kk] = __builtin_convertvector(e, V8);
            }
            if constexpr (!ONES) l_run[f] += psum;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < DVF; ++i) {
                V8 va;
                if constexpr (VTAB) va = *reinterpret_cast<const V8*>(Vs + voff2[i][hf * 2 + kk]);
                else {
                    int a = voff[hf * 2 + kk];
                    // volatile: left to itself the compiler keeps all DVF * NH * 2 XORed offsets live across the tile loop and spills (d = 160)
                    if constexpr (VSW2) { if (vx(i)) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "s"(vx(i))); }
                    va = *reinterpret_cast<const V8*>(Vs + a + i * 16 * VROW);
                }
#pragma unroll
                for (int f = 0; f < QF; ++f) oacc[i][f] = Tag::mfma16(va, pf[f][kk], oacc[i][f]);
            }
        }
    };

    const int n_tiles = (Ltot + KB2 - 1) / KB2;
    dma_k(0, 0);
    load_v(0);
    __syncthreads();          // V^T zero / ones fill complete
    store_v(0);
    __syncthreads();          // tile 0 visible (the barrier drains the LDS-DMA)

    for (int t = 0; t < n_tiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < n_tiles) { dma_k(t + 1, cur ^ 1); load_v(t + 1); }
        const unsigned char* Ks = Kbuf + cur * K_BYTES;
        const unsigned char* Vs = Vbuf + cur * V_BYTES;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            const int key0 = t * KB2 + hf * 64;
            if (key0 + 64 <= Ltot) {
                do_half(Ks, Vs, hf, key0, std::false_type{});
            } else if (key0 < Ltot) {
                do_half(Ks, Vs, hf, key0, std::true_type{});
            }
        }
        if (t + 1 < n_tiles) store_v(cur ^ 1);
        __syncthreads();
    }

    typedef T T4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l;
        if constexpr (ONES) {
            l = __shfl(oacc[DVF - 1][f][L_R], L_G * 16 + l16, 64);    // row D of O^T is sum_j P for query l16
        } else {
            l = l_run[f];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int q = q_base + f * 16 + l16;
        if (q < p.Lq) {
            T* orow = reinterpret_cast<T*>(p.O) + ((size_t)b * p.Lq + q) * p.ldo + h * D;
#pragma unroll
            for (int i = 0; i < DVF; ++i) {
                const int dv = i * 16 + g * 4;
                if (dv < D) {
                    T4 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[r] = Tag::from_f32(oacc[i][f][r] * inv);
                    *reinterpret_cast<T4*>(orow + dv) = pk;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_attention3 (d = 40, the head width of every level-0 SD-1.5 attention = 86 % of the UNet's attention FLOPs).
// The 16x16x32 kernel above pads d = 40 to 64 for Q K^T and spends a third of its VALU on transposing V through registers.  Here:
//   * S^T = K Q^T on v_mfma_f32_32x32x16: d = 40 is three 16-deep steps (48, 83 % useful instead of 62 %); a wave owns 32 queries, its lane
//     holds query (lane & 31) and 16 of the 32 keys of a block, so the row maximum needs ONE cross-lane step (v_permlane32_swap);
//   * O^T = V^T P^T stays on v_mfma_f32_16x16x32 (dv = 40 -> 48 rows; row 40 is the all-ones row that accumulates the softmax denominator).
//     P moves from the 32x32 accumulator layout into its B-operand layout with four v_permlane16_swap per 32 keys (after packing to 16 bit);
//     the MFMA's contraction slots are DEFINED by where those swaps leave the keys: lane group g of the operand holds keys
//     {0-3, 16-19} + {0, 8, 4, 12}[g] of the block;
//   * V is staged ROW-MAJOR by LDS-DMA (no register transpose, no ds_write) into 96-byte rows [40 values | 1 0 0 0 0 0 0 0] -- the last
//     chunk comes from a constant page -- and the V^T fragments are read with the LDS transpose read ds_read_b64_tr_b16: a 16-lane group
//     reads a [4 keys][16 dv] sub-matrix and each lane receives one dv column.  Key k sits in LDS row k with bits 2 and 3 swapped, so the
//     rows a half-wave touches in one read are 8 consecutive ones (24 banks apart: conflict-free);
//   * K rows are 80 bytes, unpadded: ds_read_b128 lane groups cover 16 distinct rows mod 16 and 20 banks per row generate every multiple
//     of 4 -- conflict-free without a swizzle.  The upper half of the third d-step reads chunk 4 again (finite data) against zero Q.
// 64 keys per LDS stage (11 KiB = 11 LDS-DMA instructions per block), double buffered, one barrier per stage; OCC blocks per CU.
// ---------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned short g_attn_ones_f16[8] = {0x3C00u, 0, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) unsigned short g_attn_ones_bf16[8] = {0x3F80u, 0, 0, 0, 0, 0, 0, 0};

// development aid (mve_attention_profile): with ABL bit 10 a wave adds its shader-clock (s_memtime) and 100 MHz (s_memrealtime) durations here
__device__ unsigned long long g_attn_prof[4] = {0ull, 0ull, 0ull, 0ull};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NW waves per block (32 queries each: 128 or 256 queries share one K/V stage -- the K/V bytes pulled through the LDS-DMA path per FLOP
// halve with NW = 8), NST LDS stages: 2 = the next tile is in flight during a tile, drained by __syncthreads(); 3 = two tiles in flight,
// raw s_barrier + counted s_waitcnt vmcnt(n) so that only the older tile is waited for.
// LDS-DMA issued from inline asm (one 16-byte chunk per lane, wave-uniform LDS destination in M0): hipcc does not see the LDS write, so it
// neither drains it in front of LDS reads it cannot disambiguate nor at barriers -- the kernel counts vmcnt itself.
__device__ __forceinline__ void attn_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// PRE: Q arrives multiplied by softmax_scale * log2(e) (the UNet executor folds the factor into the to_q weights when it packs them, one
// rounding either way), so the logits leave the MFMA in log2 units and the running maximum is subtracted BY the MFMA: the first d-step
// takes C = -m_run (16 registers holding the lane's own query's value) instead of 0.  exp2 is then applied to the accumulator as it is:
// the 32 v_fma per tile of the generic path disappear (rocprofv3: the kernel is VALU-issue bound -- SQ_ACTIVE_INST_VALU 77 % of the
// SIMD cycles against 38 % MFMA busy).  A growing maximum (rare after the first tiles) subtracts the growth from S and rescales O.
// ABL: timing-only ablation mask (tools/ab_attention_ablate.py; results are WRONG for ABL != 0).  bit 0: no s_barrier; 1: no vmcnt wait;
// 2: no LDS-DMA; 3: no v_exp; 4: no row-maximum chain; 5: no Q K^T MFMAs; 6: no P V MFMAs; 7: no K fragment reads; 8: no V^T fragment reads;
// 9: no permlane16 swaps
template <class Tag, bool SEG2, int NW, int NST, int WPS, bool PRE, bool PIPE, int ABL = 0>
__global__ __launch_bounds__(64 * NW, WPS) void k_attention3(const AttnParams p) {
    constexpr int D = 40, KB = 64, QB = 32 * NW;
    constexpr int K_ROW = 80, V_ROW = 96;
    constexpr int K_BYTES = KB * K_ROW, V_BYTES = KB * V_ROW, STAGE = K_BYTES + V_BYTES;
    constexpr int N_DMA = STAGE / 1024, K_DMA = K_BYTES / 1024, DMA_PER_WAVE = (N_DMA + NW - 1) / NW;
    static_assert(STAGE % 1024 == 0 && K_BYTES % 1024 == 0, "stage must split into whole LDS-DMA instructions");
    static_assert(NST >= 2 && NST <= 4, "2 to 4 LDS stages");
    typedef typename Tag::V8 V8;
    typedef typename Tag::T T;
    typedef T T2 __attribute__((ext_vector_type(2)));
    typedef T T4 __attribute__((ext_vector_type(4)));

    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l32 = lane & 31, hi = lane >> 5, l16 = lane & 15, g = lane >> 4;
    unsigned long long prof_t0 = 0, prof_r0 = 0;
    if constexpr ((ABL & 1024) != 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t0), "=s"(prof_r0) :: "memory");

    const int q_tiles = (p.Lq + QB - 1) / QB;
    const unsigned nblk = (unsigned)(q_tiles * p.heads * p.B);
    const unsigned bid = mve_xcd_remap(blockIdx.x, nblk);
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % p.heads;
    const int b = bid / (q_tiles * p.heads);
    const int Ltot = p.Lk + p.Lk2;

    const T* kb1 = reinterpret_cast<const T*>(p.K) + (size_t)b * p.Lk * p.ldk + h * D;
    const T* vb1 = reinterpret_cast<const T*>(p.V) + (size_t)b * p.Lk * p.ldv + h * D;
    const T* kb2 = SEG2 ? reinterpret_cast<const T*>(p.K2) + (size_t)b * p.Lk2 * p.ldk2 + h * D : nullptr;
    const T* vb2 = SEG2 ? reinterpret_cast<const T*>(p.V2) + (size_t)b * p.Lk2 * p.ldv2 + h * D : nullptr;
    const T* ones = reinterpret_cast<const T*>(std::is_same<T, f16>::value ? g_attn_ones_f16 : g_attn_ones_bf16);
    auto k_row = [&](int j) -> const T* {
        if constexpr (SEG2) { if (j >= p.Lk) return kb2 + (size_t)(j - p.Lk) * p.ldk2; }
        return kb1 + (size_t)j * p.ldk;
    };
    auto v_row = [&](int j) -> const T* {
        if constexpr (SEG2) { if (j >= p.Lk) return vb2 + (size_t)(j - p.Lk) * p.ldv2; }
        return vb1 + (size_t)j * p.ldv;
    };

    // Q fragments (B operand of the 32x32x16 MFMA): lane = (query l32, d half hi), step s covers d = 16 s + 8 hi .. + 7
    const int q_base = qt * QB + wid * 32;
    V8 qf[3];
    {
        int q = q_base + l32;
        q = q < p.Lq ? q : p.Lq - 1;
        const T* row = reinterpret_cast<const T*>(p.Q) + ((size_t)b * p.Lq + q) * p.ldq + h * D;
        qf[0] = __builtin_bit_cast(V8, *reinterpret_cast<const u32x4*>(row + 8 * hi));
        qf[1] = __builtin_bit_cast(V8, *reinterpret_cast<const u32x4*>(row + 16 + 8 * hi));
        u32x4 t = {0u, 0u, 0u, 0u};
        if (hi == 0) t = *reinterpret_cast<const u32x4*>(row + 32);
        qf[2] = __builtin_bit_cast(V8, t);
    }

    // LDS-DMA slots of this lane: instruction `inst` = wv + NW i writes stage bytes [1024 inst, 1024 inst + 1024), 16 per lane.
    // d_src[i] = source of the lane's chunk for key row 0 of a tile, d_ld[i] = elements per key row (0 for the constant chunk), so that the
    // source for tile t is d_src + t * 64 * d_ld: one 64-bit add per instruction and tile (single KV segment, full tiles).  The general
    // path (second KV segment, clamped rows of the last tile) recomputes the row from d_key / d_col.
    const int wv = __builtin_amdgcn_readfirstlane(wid);            // provably wave-uniform: scalar branches around the per-wave DMA slots
    const unsigned smem_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    int d_key[DMA_PER_WAVE], d_col[DMA_PER_WAVE];
    const T* d_src[DMA_PER_WAVE];
    int d_ld[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int inst = wv + NW * i;
        const int o = inst * 1024 + lane * 16;
        if (inst < K_DMA) {
            const int c = o >> 4;
            d_key[i] = c / 5; d_col[i] = (c - d_key[i] * 5) * 8;
            d_ld[i] = p.ldk;
            d_src[i] = kb1 + (size_t)d_key[i] * p.ldk + d_col[i];
        } else {
            const int c = (o - K_BYTES) >> 4;
            const int r = c / 6, col = c - r * 6;
            d_key[i] = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);      // LDS row r holds key r with bits 2 and 3 swapped
            d_col[i] = col < 5 ? col * 8 : -1;                            // chunk 5: the constant [1 0 0 0 0 0 0 0]
            d_ld[i] = col < 5 ? p.ldv : 0;
            d_src[i] = col < 5 ? vb1 + (size_t)d_key[i] * p.ldv + d_col[i] : ones;
        }
    }
    // full tile of a single-segment problem: running pointers
    auto dma_fast = [&](int stage_off) {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int inst = wv + NW * i;
            if (inst < N_DMA) {
                attn_dma16(d_src[i], smem_base + stage_off + inst * 1024);
                d_src[i] += (size_t)KB * d_ld[i];
            }
        }
    };
    auto dma_any = [&](int t, int stage_off) {
        const int j0 = t * KB;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int inst = wv + NW * i;
            if (inst < N_DMA) {
                int j = j0 + d_key[i];
                j = j < Ltot ? j : Ltot - 1;                  // clamped rows hold finite data; their scores are masked
                const T* src = inst < K_DMA ? k_row(j) + d_col[i] : (d_col[i] >= 0 ? v_row(j) + d_col[i] : ones);
                attn_dma16(src, smem_base + stage_off + inst * 1024);
            }
        }
    };
    // tile t (t >= 1 is always issued one tile ahead, in order)
    const int n_tiles = (Ltot + KB - 1) / KB, n_full = Ltot / KB;
    auto dma = [&](int t, int stage_off) {
        if constexpr ((ABL & 4) != 0) return;
        if constexpr (SEG2) dma_any(t, stage_off);
        else { if (t < n_full) dma_fast(stage_off); else dma_any(t, stage_off); }
    };
    // number of LDS-DMA instructions this wave issues per tile (wave-uniform): what a counted vmcnt leaves in flight
    const int n_mine = (N_DMA - wv + NW - 1) / NW;

    // fragment read offsets (lane constants)
    const int k_off01 = l32 * K_ROW + hi * 16;                  // d-steps 0, 1: + 32 s
    const int k_off2 = l32 * K_ROW + 64;                        // d-step 2: chunk 4 for both halves (Q is zero in the upper one)
    const int v_off = K_BYTES + (lane >> 2) * V_ROW + (lane & 3) * 8;

    f32x4 oacc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f) oacc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = PRE ? 0.f : -INFINITY;
    const float sc = p.scale_log2e;

    // PRE: -m_run in all 16 accumulator registers (C operand of the first d-step)
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    struct S2 { f32x16 a, b; };                    // S^T of one 64-key tile: key blocks 0-31 and 32-63
    struct PB { unsigned w[2][2][4]; };            // P^T in the PV operand layout: [key block][16-query group][register]

    // S^T = K Q^T of the tile staged at St (PRE: minus the running maximum through the C operand)
    auto qk = [&](const unsigned char* St) -> S2 {
        f32x16 zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
        S2 s;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int o = st < 2 ? k_off01 + 32 * st : k_off2;
            V8 ka, kb_;
            if constexpr ((ABL & 128) != 0) { ka = qf[st]; kb_ = qf[(st + 1) % 3]; }
            else { ka = *reinterpret_cast<const V8*>(St + o); kb_ = *reinterpret_cast<const V8*>(St + o + 32 * K_ROW); }
            if constexpr ((ABL & 32) != 0) {          // no MFMA: keep the fragments alive, S = C operand
                asm volatile("" :: "v"(ka), "v"(kb_));
                if (st == 0) { s.a = negm; s.b = negm; }
                continue;
            }
            if (PRE && st == 0) {
                // D != C written out by hand: for a C operand that stays live hipcc copies it into the accumulator first (8 v_mov_b64 per
                // chain and tile -- what the saved v_fma bought).  Operands come from waited-for loads / old VALU results; the only
                // consumer of D is the next MFMA of the chain taking it whole as C: no wait states needed on either side.
                if constexpr (std::is_same<T, f16>::value) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s.a) : "v"(ka), "v"(qf[0]), "v"(negm));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s.b) : "v"(kb_), "v"(qf[0]), "v"(negm));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s.a) : "v"(ka), "v"(qf[0]), "v"(negm));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s.b) : "v"(kb_), "v"(qf[0]), "v"(negm));
                }
            } else {
                s.a = Tag::mfma32(ka, qf[st], st == 0 ? zero16 : s.a);
                s.b = Tag::mfma32(kb_, qf[st], st == 0 ? zero16 : s.b);
            }
        }
        return s;
    };

    // online softmax of one tile, first half: mask the ragged tail, update m_run where the maximum grows (exact rescale of O; PRE: the
    // logits already carry -m_run, so growth shows as a positive maximum and is subtracted from S)
    auto softmax_max = [&](S2& s, int key0, auto masked_tag, bool first) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        if constexpr (MASKED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Ltot) s.a[r] = -INFINITY;
                if (key0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Ltot) s.b[r] = -INFINITY;
            }
        }
        float mx = fmaxf(s.a[0], s.b[0]);
        if constexpr ((ABL & 16) == 0) {
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s.a[r]), s.b[r]);      // v_max3_f32 (built with -fno-honor-nans: no canonicalising v_max)
            mx = mve_max_xor32(mx);
        }
        float a0 = 1.f, a1 = 1.f;
        bool rescale = false;
        if constexpr (PRE) {
            // s = logit - m_run (m_run = 0 before the first tile): the maximum grows where mx > 0
            if (__builtin_expect(first || __any(mx > 0.f), 0)) {
                const float delta = first ? mx : fmaxf(mx, 0.f);
                m_run += delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s.a[r] -= delta; s.b[r] -= delta; negm[r] = -m_run; }
                if (!first) {                            // O is still zero on the first tile (and exp2(-delta) may overflow there)
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    const auto ar = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
                    a0 = __uint_as_float(ar[0]); a1 = __uint_as_float(ar[1]);
                    rescale = true;
                }
            }
        } else {
            const float mxs = mx * sc;
            if (__builtin_expect(__any(mxs > m_run), 0)) {   // some query's running maximum grows: rescale (exact); rare after the first tiles
                const float m_new = fmaxf(m_run, mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                // O^T's lane column is query (lane & 15) of its 16-query group: alpha of lanes {0-15, 32-47} resp. {16-31, 48-63}
                const auto ar = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
                a0 = __uint_as_float(ar[0]); a1 = __uint_as_float(ar[1]);
                rescale = true;
            }
        }
        if (rescale) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { oacc[i][0][r] *= a0; oacc[i][1][r] *= a1; }
        }
    };

    // second half (straight-line): P = exp2(S - m), rounded to 16 bit, moved into the PV operand layout
    auto softmax_exp = [&](const S2& s) -> PB {
        const float nm = -m_run;
        PB pb;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f32x16& sk = kb ? s.b : s.a;
            unsigned pk[8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                float e0, e1;
                if constexpr (PRE && (ABL & 8) != 0) {
                    e0 = sk[2 * r2]; e1 = sk[2 * r2 + 1];
                } else if constexpr (PRE) {
                    e0 = __builtin_amdgcn_exp2f(sk[2 * r2]);
                    e1 = __builtin_amdgcn_exp2f(sk[2 * r2 + 1]);
                } else {
                    e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sk[2 * r2], sc, nm));
                    e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sk[2 * r2 + 1], sc, nm));
                }
                pk[r2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{e0, e1}, T2));
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ai = v < 2 ? v : v + 2;
                if constexpr ((ABL & 512) != 0) { pb.w[kb][0][v] = pk[ai]; pb.w[kb][1][v] = pk[ai + 2]; continue; }
                const auto r = __builtin_amdgcn_permlane16_swap(pk[ai], pk[ai + 2], false, false);
                pb.w[kb][0][v] = r[0];
                pb.w[kb][1][v] = r[1];
            }
        }
        return pb;
    };

    // O^T += V^T P^T.  V^T fragments by LDS transpose reads, issued from inline asm with their own lgkmcnt wait: hipcc has no memory operand
    // for the transpose-read builtin.  In-order LDS returns make the extra entries on the counter harmless for the compiler's own counted
    // waits (they can only over-wait).
    auto pv = [&](const unsigned char* St, const PB& pb) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s16x4 lo[3], up[3];
            const unsigned va0 = (unsigned)(uintptr_t)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW);
            if constexpr ((ABL & 256) != 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { lo[i] = __builtin_bit_cast(s16x4, u32x2{pb.w[kb][0][i], va0}); up[i] = __builtin_bit_cast(s16x4, u32x2{va0, pb.w[kb][1][i]}); }
            } else {
                // the transpose-read builtin: the compiler counts lgkmcnt itself and lets the first MFMAs start while the later reads are still
                // in flight (the round-2 asm block waited for all six: +3..4 % on the level-0 self-attention, tools/attn_lab)
                typedef __attribute__((address_space(3))) s16x4* lp;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    lo[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW + 32 * i));
                    up[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW + 32 * i + 16 * V_ROW));
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const V8 va = __builtin_bit_cast(V8, __builtin_shufflevector(lo[i], up[i], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const u32x4 pw = {pb.w[kb][f][0], pb.w[kb][f][1], pb.w[kb][f][2], pb.w[kb][f][3]};
                    if constexpr ((ABL & 64) != 0) { asm volatile("" : "+v"(oacc[i][f]) : "v"(va), "v"(pw)); continue; }
                    oacc[i][f] = Tag::mfma16(va, __builtin_bit_cast(V8, pw), oacc[i][f]);
                }
            }
        }
    };

    auto wait_sync = [&](int keep) {          // s_waitcnt vmcnt(keep) (wave-uniform, 0..3), drain the LDS reads, barrier
        if constexpr ((ABL & 3) == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }
        else if constexpr ((ABL & 1) != 0) { asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); return; }
        else if constexpr ((ABL & 2) != 0) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); return; }
        if (keep >= 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (keep == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (keep == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto next_stage = [&](int off) { return off + STAGE == NST * STAGE ? 0 : off + STAGE; };

    if constexpr (!PIPE) {
        // NST stages, PD = NST - 1 tiles in flight.  Iteration t: issue tile t + PD into the stage tile t - 1 left (every wave is past the
        // barrier that ended iteration t - 1), compute tile t, wait until only this wave's instructions of the tiles AFTER t + 1 are
        // outstanding (VMEM returns in order: tile t + 1 has landed), drain the LDS reads, barrier.
        constexpr int PD = NST - 1;
        static_assert(DMA_PER_WAVE * (PD - 1) <= 3, "wait_sync covers at most three outstanding instructions");
#pragma unroll
        for (int i = 0; i < PD; ++i)
            if (i < n_tiles) dma(i, i * STAGE);
        // make hipcc wait for the Q loads HERE: left alone it puts their s_waitcnt vmcnt(0) at the first use inside the tile loop, where it
        // would drain the in-flight LDS-DMA of every iteration
        asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]));
        wait_sync(n_mine * ((n_tiles < PD ? n_tiles : PD) - 1));
        int cur = 0, nxt = PD * STAGE;            // stage offsets of tile t and of tile t + PD
        for (int t = 0; t < n_full; ++t) {        // full tiles: no key mask
            if (t + PD < n_tiles) dma(t + PD, nxt);
            S2 s = qk(smem + cur);
            softmax_max(s, t * KB, std::false_type{}, t == 0);
            const PB pb = softmax_exp(s);
            pv(smem + cur, pb);
            const int last_issued = t + PD < n_tiles ? t + PD : n_tiles - 1;
            wait_sync(last_issued > t + 1 ? n_mine * (last_issued - (t + 1)) : 0);
            nxt = cur;
            cur = next_stage(cur);
        }
        if (n_full < n_tiles) {                   // ragged last tile (Lk = 77, ...): already landed
            S2 s = qk(smem + cur);
            softmax_max(s, n_full * KB, std::true_type{}, n_full == 0);
            const PB pb = softmax_exp(s);
            pv(smem + cur, pb);
        }
    } else {
        // Software pipeline inside the wave: while the VALU runs the exponentials of tile t, the matrix pipe already computes S^T of tile t + 1
        // (two independent instruction streams in one basic block, after the branchy maximum update so that the new tile's C operand is the
        // updated maximum; rocprofv3 showed the un-pipelined wave spending 46 % of its cycles
        // issue-stalled behind its own MFMA -> max -> exp -> MFMA chain with neither pipe saturated).  K of tile t + 1 must therefore be in
        // LDS during iteration t: tile t + 2 is issued at the top of iteration t and fully waited for at its end (NST = 3: it lands in the
        // stage tile t - 1 left; NST = 4: tile t + 3 is issued and one tile stays in flight across the barrier).
        static_assert(NST >= 3, "the pipelined loop needs the next tile's K resident");
        constexpr int PD = NST - 1;
        static_assert(DMA_PER_WAVE * (PD - 2) <= 3, "wait_sync covers at most three outstanding instructions");
#pragma unroll
        for (int i = 0; i < PD; ++i)
            if (i < n_tiles) dma(i, i * STAGE);
        asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]));
        {   // tiles 0 and 1 landed (the second one only if it exists); later ones may stay in flight
            const int issued = n_tiles < PD ? n_tiles : PD;
            wait_sync(issued > 2 ? n_mine * (issued - 2) : 0);
        }
        int cur = 0, nx1 = STAGE, nxt = (PD % NST) * STAGE;      // stage offsets of tiles t, t + 1 and t + PD
        S2 sA = qk(smem), sB;
        const int n_main = n_tiles - 1;           // tiles that have a successor; all of them are full tiles
        auto step = [&](S2& sc_, S2& sn_, int t) {
            if (t + PD < n_tiles) dma(t + PD, nxt);
            softmax_max(sc_, t * KB, std::false_type{}, t == 0);           // branchy part first: m_run (and PRE's C operand) final for tile t
            sn_ = qk(smem + nx1);                                          // matrix pipe: S^T of tile t + 1 ...
            const PB pb = softmax_exp(sc_);                                // ... under the VALU's exponentials of tile t
            pv(smem + cur, pb);
            // tiles up to t + 2 must have landed before the next iteration reads K of tile t + 2
            const int last_issued = t + PD < n_tiles ? t + PD : n_tiles - 1;
            wait_sync(last_issued > t + 2 ? n_mine * (last_issued - (t + 2)) : 0);
            nxt = cur; cur = nx1; nx1 = next_stage(nx1);
        };
        int t = 0;
        for (; t + 2 <= n_main; t += 2) { step(sA, sB, t); step(sB, sA, t + 1); }
        if (t < n_main) { step(sA, sB, t); sA = sB; ++t; }
        if (n_full < n_tiles) softmax_max(sA, t * KB, std::true_type{}, t == 0);
        else softmax_max(sA, t * KB, std::false_type{}, t == 0);
        const PB pb = softmax_exp(sA);
        pv(smem + cur, pb);
    }

#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float l = __shfl(oacc[2][f][0], 32 + l16, 64);       // row dv = 40 of O^T (fragment 2, lane group 2, register 0) = sum_j P
        const float inv = 1.0f / l;
        const int q = q_base + f * 16 + l16;
        if (q < p.Lq) {
            T* orow = reinterpret_cast<T*>(p.O) + ((size_t)b * p.Lq + q) * p.ldo + h * D;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int dv = i * 16 + g * 4;
                if (dv < D) {
                    T4 o4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o4[r] = Tag::from_f32(oacc[i][f][r] * inv);
                    *reinterpret_cast<T4*>(orow + dv) = o4;
                }
            }
        }
    }
    if constexpr ((ABL & 1024) != 0) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
        if (lane == 0 && (blockIdx.x & 63) == 5) {        // a sample of the blocks: contended atomics cost ~12 ns each
            atomicAdd(&g_attn_prof[0], t1 - prof_t0);
            atomicAdd(&g_attn_prof[1], r1 - prof_r0);
            atomicAdd(&g_attn_prof[2], 1ull);
        }
    }
}

// 0: the measured configuration.  2 (d = 80 / 160, single KV segment): conflict-free K swizzle, same arithmetic (bit-identical results).
// 4 (single KV segment): V^T store swizzle VSW2, same arithmetic (bit-identical results); 6 = 2 + 4.
// 1 (d = 40 only, single KV segment): 16 query rows per wave, 64-key fills, 3 blocks per CU --
// an experiment for the VALU-bound d = 40 case (mve_attention_tune; results are NOT bit-identical across variants: the online-softmax
// rescale points move with the fill size).
int g_attn_variant = 9;
int g_attn_ablate = 0;       // k_attention3's ABL mask (timing experiments only; mve_attention_tune bits 8-19)

template <class Tag, int D>
int launch(const AttnParams& p, hipStream_t s) {
    if constexpr (D == 40) {
        if (g_attn_variant >= 8 && g_attn_variant <= 15) {                        // k_attention3: 32x32x16 Q K^T, LDS-transposed V
            // 8: 4 waves, 2 stages; 9: 8 waves, 2 stages; 10: 4 waves, 3 stages; 11: 8 waves, 3 stages;
            // 12..15: the software-pipelined loop (S^T of tile t + 1 under the softmax of tile t), 3 waves per SIMD:
            //   12: 4 waves, 3 stages; 13: 4 waves, 4 stages; 14: 8 waves, 3 stages (2 per SIMD); 15: 8 waves, 4 stages
            const int var = g_attn_variant;
            const bool w8 = var == 9 || var == 11 || var >= 14;
            const unsigned grid3 = (unsigned)(((p.Lq + (w8 ? 255 : 127)) / (w8 ? 256 : 128)) * p.heads * p.B);
#define MVE_A3P(SEG, NW_, NST_, WPS_, PRE_, PIPE_) k_attention3<Tag, SEG, NW_, NST_, WPS_, PRE_, PIPE_><<<grid3, 64 * NW_, 0, s>>>(p)
            // PRE (maximum subtracted through the MFMA's C operand) is compiled for the un-pipelined variants only: same-box A/B showed no gain
            // from the 32 fewer v_fma per tile (the loop is bound by its dependency chain, not by VALU issue), and its 16 extra registers
            // spill in the pipelined loop.  A pre-scaled Q runs the generic path with scale_log2e = 1.
#define MVE_A3(NW_, NST_, WPS_, PIPE_) do { \
                if (p.Lk2 > 0) { if (p.prescaled && !PIPE_) MVE_A3P(true, NW_, NST_, WPS_, (!PIPE_), PIPE_); else MVE_A3P(true, NW_, NST_, WPS_, false, PIPE_); } \
                else { if (p.prescaled && !PIPE_) MVE_A3P(false, NW_, NST_, WPS_, (!PIPE_), PIPE_); else MVE_A3P(false, NW_, NST_, WPS_, false, PIPE_); } } while (0)
#ifdef MVE_ATTN_LAB                 // (development builds only: `python -m mvedit_amd.build` with MVE_ATTN_LAB=1 in the environment)
            if (g_attn_ablate != 0) {       // timing-only ablations of the default configuration (fp16, single segment, pre-scaled Q)
                if constexpr (std::is_same<Tag, F16Tag>::value) {
                    if (p.Lk2 == 0 && p.prescaled) {
                        const unsigned grid9 = (unsigned)(((p.Lq + 255) / 256) * p.heads * p.B);
                        bool hit = true;
                        switch (g_attn_ablate) {
#define MVE_ABL(A) case A: k_attention3<Tag, false, 8, 2, 4, true, false, (A) | 1024><<<grid9, 512, 0, s>>>(p); break;
                            MVE_ABL(2048) MVE_ABL(1) MVE_ABL(3) MVE_ABL(7) MVE_ABL(8) MVE_ABL(16) MVE_ABL(24) MVE_ABL(32) MVE_ABL(64) MVE_ABL(96) MVE_ABL(128) MVE_ABL(256)
                            MVE_ABL(384) MVE_ABL(512) MVE_ABL(536) MVE_ABL(927) MVE_ABL(480) MVE_ABL(1023)
#undef MVE_ABL
                            default: hit = false;
                        }
                        if (hit) { MVE_LAUNCH_CHECK(); return MVE_OK; }
                    }
                }
            }
#endif
            switch (var) {
                case 8: MVE_A3(4, 2, 4, false); break;
                case 9: MVE_A3(8, 2, 4, false); break;
                case 10: MVE_A3(4, 3, 4, false); break;
                case 11: MVE_A3(8, 3, 4, false); break;
                case 12: MVE_A3(4, 3, 3, true); break;
                case 13: MVE_A3(4, 4, 3, true); break;
                case 14: MVE_A3(8, 3, 2, true); break;
                default: MVE_A3(8, 4, 2, true); break;
            }
#undef MVE_A3P
#undef MVE_A3
            MVE_LAUNCH_CHECK();
            return MVE_OK;
        }
        if (g_attn_variant == 1 && p.Lk2 == 0) {
            const unsigned grid1 = (unsigned)(((p.Lq + 63) / 64) * p.heads * p.B);
            k_attention2<Tag, D, false, 1, 64, 3><<<grid1, NT, 0, s>>>(p);
            MVE_LAUNCH_CHECK();
            return MVE_OK;
        }
    }
    constexpr int QB = 128;
    const unsigned grid = (unsigned)(((p.Lq + QB - 1) / QB) * p.heads * p.B);
    // the default family (variants >= 8) uses, for the other head dims, the LDS layouts that the same-box A/B measured fastest and the GPU
    // tests showed bit-identical to variant 0: K + V^T swizzles for d = 80 / 160 (+5 %), the V^T store swizzle for d = 64 (+12 %)
    const int var = g_attn_variant >= 8 ? ((D == 80 || D == 160) ? 6 : 4) : g_attn_variant;
    if constexpr (D == 80 || D == 160) {
        if ((var == 2 || var == 6) && p.Lk2 == 0) {          // conflict-free K swizzle (see k_perm)
            if (var == 6) k_attention2<Tag, D, false, 2, 0, 0, true, true><<<grid, NT, 0, s>>>(p);
            else k_attention2<Tag, D, false, 2, 0, 0, true><<<grid, NT, 0, s>>>(p);
            MVE_LAUNCH_CHECK();
            return MVE_OK;
        }
    }
    if ((var == 4 || var == 6) && p.Lk2 == 0) {              // V^T store swizzle (see v_swz2)
        k_attention2<Tag, D, false, 2, 0, 0, false, true><<<grid, NT, 0, s>>>(p);
        MVE_LAUNCH_CHECK();
        return MVE_OK;
    }
    if (p.Lk2 > 0) k_attention2<Tag, D, true><<<grid, NT, 0, s>>>(p);
    else k_attention2<Tag, D, false><<<grid, NT, 0, s>>>(p);
    MVE_LAUNCH_CHECK();
    return MVE_OK;
}

template <class Tag>
int dispatch_d(const AttnParams& p, int d, hipStream_t s) {
    switch (d) {
        case 40: return launch<Tag, 40>(p, s);
        case 64: return launch<Tag, 64>(p, s);
        case 80: return launch<Tag, 80>(p, s);
        case 160: return launch<Tag, 160>(p, s);
        default:
            mve_set_error("attention: unsupported head dim %d (supported: 40, 64, 80, 160)", d);
            return MVE_ERR_ARG;
    }
}

}  // namespace

// development aid: sums over the waves of the profiled (ablation) launches since the last call: {shader cycles, 10 ns ticks, waves, 0}; resets them
extern "C" int mve_attention_profile(unsigned long long* out4) {
    MVE_CHECK(out4, MVE_ERR_ARG, "attention_profile: null pointer");
#ifndef MVE_ATTN_LAB
    mve_set_error("attention_profile: the timing-only ablation kernels are compiled into development builds only (MVE_ATTN_LAB=1 python -m mvedit_amd.build)");
    return MVE_ERR_STATE;
#endif
    MVE_HIP(hipDeviceSynchronize());
    MVE_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_attn_prof), 4 * sizeof(unsigned long long)));
    const unsigned long long z[4] = {0ull, 0ull, 0ull, 0ull};
    MVE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), z, sizeof(z)));
    return MVE_OK;
}

extern "C" int mve_attention_tune(int variant) {
    const int old = g_attn_variant | (g_attn_ablate << 8);
#ifndef MVE_ATTN_LAB
    // release builds carry no ablation kernels: a word with bits 8-19 set would otherwise select nothing silently
    if (variant >= 256) { mve_set_error("attention_tune: ablation bits (8-19) need a development build (MVE_ATTN_LAB=1)"); return MVE_ERR_ARG; }
#endif
    if (variant >= 0) { g_attn_variant = variant & 255; g_attn_ablate = (variant >> 8) & 4095; }
    return old;
}

static int attention_entry(int dtype, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                           const void* K2, int ldk2, const void* V2, int ldv2, void* O, int ldo, int B, int Lq,
                           int Lk, int Lk2, int heads, int head_dim, float scale, int prescaled, void* stream) {
    if (B == 0 || Lq == 0) return MVE_OK;
    MVE_CHECK(Q && K && V && O, MVE_ERR_ARG, "attention: null pointer");
    MVE_CHECK(Lk > 0 && Lk2 >= 0 && heads > 0, MVE_ERR_ARG, "attention: bad sizes Lk=%d Lk2=%d heads=%d", Lk, Lk2, heads);
    MVE_CHECK(Lk2 == 0 || (K2 && V2), MVE_ERR_ARG, "attention: second KV segment has null pointers");
    MVE_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && (Lk2 == 0 || (ldk2 % 8 == 0 && ldv2 % 8 == 0)),
              MVE_ERR_ARG, "attention: leading dimensions must be multiples of 8");
    AttnParams p;
    p.Q = Q; p.K = K; p.V = V; p.K2 = K2; p.V2 = V2; p.O = O;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldk2 = ldk2; p.ldv2 = ldv2; p.ldo = ldo;
    p.B = B; p.Lq = Lq; p.Lk = Lk; p.Lk2 = Lk2; p.heads = heads;
    p.scale_log2e = prescaled ? 1.0f : scale * 1.4426950408889634f;
    p.prescaled = prescaled;
    if (dtype == MVE_F16) return dispatch_d<F16Tag>(p, head_dim, (hipStream_t)stream);
    if (dtype == MVE_BF16) return dispatch_d<BF16Tag>(p, head_dim, (hipStream_t)stream);
    mve_set_error("attention: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}

extern "C" int mve_attention(int dtype, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                             const void* K2, int ldk2, const void* V2, int ldv2, void* O, int ldo, int B, int Lq,
                             int Lk, int Lk2, int heads, int head_dim, float scale, void* stream) {
    return attention_entry(dtype, Q, ldq, K, ldk, V, ldv, K2, ldk2, V2, ldv2, O, ldo, B, Lq, Lk, Lk2, heads, head_dim, scale, 0, stream);
}

extern "C" int mve_attention_prescaled(int dtype, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                       const void* K2, int ldk2, const void* V2, int ldv2, void* O, int ldo, int B, int Lq,
                                       int Lk, int Lk2, int heads, int head_dim, void* stream) {
    return attention_entry(dtype, Q, ldq, K, ldk, V, ldv, K2, ldk2, V2, ldv2, O, ldo, B, Lq, Lk, Lk2, heads, head_dim, 1.0f, 1, stream);
}
