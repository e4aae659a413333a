// Attention lab (development aid, not part of the product): fast-iteration harness for the head_dim 40 flash-attention loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -I include tools/attn_lab/attn_lab.hip -o tools/attn_lab/attn_lab
// Builds kernel variants side by side, checks each against a naive fp32 kernel on a small problem and times it on the benchmark
// shape (B = 64 images, L = 4096 tokens, 8 heads of 40), printing per-wave shader cycles and the clock next to the wall time.
// Conventions (fragment layouts, LDS images, swaps) are those of csrc/attention.hip's k_attention3; Q arrives pre-scaled by
// head_dim^-1/2 * log2(e).
#include "../../mvedit_amd/csrc/common.h"

#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <vector>

void mve_set_error(const char*, ...) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f16 T;
typedef f16x8 V8;
typedef T T2 __attribute__((ext_vector_type(2)));
typedef T T4 __attribute__((ext_vector_type(4)));

struct AttnParams {
    const void* Q; const void* K; const void* V; void* O;
    int ldq, ldk, ldv, ldo;
    int B, Lq, Lk, heads;
    unsigned long long* prof;      // [3]: cycles, 10 ns ticks, waves (sampled blocks)
};

__device__ __attribute__((aligned(16))) unsigned short g_ones_f16[8] = {0x3C00u, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void attn_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ---- naive reference: one thread per (b, h, q), fp32 ------------------------------------------------------------------------------
__global__ void k_ref(const AttnParams p, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = p.B * p.heads * p.Lq;
    if (idx >= total) return;
    const int q = idx % p.Lq, h = (idx / p.Lq) % p.heads, b = idx / (p.Lq * p.heads);
    const T* qr = reinterpret_cast<const T*>(p.Q) + ((size_t)b * p.Lq + q) * p.ldq + h * 40;
    float qv[40];
    for (int d = 0; d < 40; ++d) qv[d] = (float)qr[d];
    float m = -INFINITY, l = 0.f, o[40];
    for (int d = 0; d < 40; ++d) o[d] = 0.f;
    for (int j = 0; j < p.Lk; ++j) {
        const T* kr = reinterpret_cast<const T*>(p.K) + ((size_t)b * p.Lk + j) * p.ldk + h * 40;
        const T* vr = reinterpret_cast<const T*>(p.V) + ((size_t)b * p.Lk + j) * p.ldv + h * 40;
        float s = 0.f;
        for (int d = 0; d < 40; ++d) s += qv[d] * (float)kr[d];
        const float mn = fmaxf(m, s);
        const float a = exp2f(m - mn), e = exp2f(s - mn);
        l = l * a + e;
        for (int d = 0; d < 40; ++d) o[d] = o[d] * a + e * (float)vr[d];
        m = mn;
    }
    for (int d = 0; d < 40; ++d) out[(size_t)idx * 40 + d] = o[d] / l;
}

// ---- kernel under development --------------------------------------------------------------------------------------------------------
// VAR bits: 1 = software pipeline (S^T of tile t + 1 under the softmax of tile t) with early fragment reads; 2 = tree-shaped row maximum;
//           4 = V^T fragments through the transpose-read builtin (compiler-counted lgkmcnt) instead of the asm block
//           32 (round 5) = two 64-key tiles per LDS stage: one barrier per 128 keys (45 KB of LDS per block)
//           8 / 16 / 24 (round 5) = the exponentials of one pair of columns in 4 / in 2 / of every pair through exp2_poly_pair instead of v_exp_f32
// 2^x for a pair of columns without the transcendental unit: clamp, round-to-nearest split by the 1.5 * 2^23 constant (the integer part lands in the
// low mantissa bits), degree-3 polynomial of the fraction on [-1/2, 1/2] (relative error 7.6e-5, under fp16's half ulp of 4.9e-4) on v_pk_fma_f32, the
// integer part shifted into the exponent field.  10 VALU instructions per pair (2 v_max, 3 v_pk_add, 3 v_pk_fma, 2 v_lshl_add) against 2 v_exp_f32.
__device__ __forceinline__ f32x2 exp2_poly_pair(float x0, float x1) {
    const f32x2 magic = {12582912.f, 12582912.f};
    const f32x2 x = {fmaxf(x0, -126.f), fmaxf(x1, -126.f)};
    const f32x2 t = x + magic;
    const f32x2 f = x - (t - magic);
    const f32x2 c3 = {0.05520550534f, 0.05520550534f}, c2 = {0.24261397123f, 0.24261397123f}, c1 = {0.69325476885f, 0.69325476885f},
                c0 = {0.99992769957f, 0.99992769957f};
    f32x2 q = __builtin_elementwise_fma(c3, f, c2);
    q = __builtin_elementwise_fma(q, f, c1);
    q = __builtin_elementwise_fma(q, f, c0);
    f32x2 r;
    r[0] = __uint_as_float(__float_as_uint(q[0]) + (__float_as_uint(t[0]) << 23));
    r[1] = __uint_as_float(__float_as_uint(q[1]) + (__float_as_uint(t[1]) << 23));
    return r;
}

template <int NW, int NST, int WPS, int VAR>
__global__ __launch_bounds__(64 * NW, WPS) void k_attn4(const AttnParams p) {
    constexpr int D = 40, KB = 64, QB = 32 * NW;
    constexpr int K_ROW = 80, V_ROW = 96;
    constexpr int SUB = (VAR & 32) ? 2 : 1;             // key tiles per LDS stage = per barrier
    constexpr int K_BYTES = KB * K_ROW, V_BYTES = KB * V_ROW, STAGE1 = K_BYTES + V_BYTES, STAGE = SUB * STAGE1;
    constexpr int N_DMA1 = STAGE1 / 1024, N_DMA = SUB * N_DMA1, K_DMA = K_BYTES / 1024, DMA_PER_WAVE = (N_DMA + NW - 1) / NW;
    static_assert(SUB == 1 || ((VAR & 1) == 0 && NW == 8), "two tiles per stage: plain loop, 8-wave blocks (3 LDS-DMAs per wave and stage)");
    constexpr bool PIPE = (VAR & 1) != 0, TREE = (VAR & 2) != 0, TRB = (VAR & 4) != 0;
    static_assert(!PIPE || NST >= 3, "the pipelined loop keeps the next tile's K resident");

    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l32 = lane & 31, hi = lane >> 5, l16 = lane & 15, g = lane >> 4;
    unsigned long long prof_t0 = 0, prof_r0 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t0), "=s"(prof_r0) :: "memory");

    const int q_tiles = (p.Lq + QB - 1) / QB;
    const unsigned nblk = (unsigned)(q_tiles * p.heads * p.B);
    const unsigned bid = mve_xcd_remap(blockIdx.x, nblk);
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % p.heads;
    const int b = bid / (q_tiles * p.heads);

    const T* kb1 = reinterpret_cast<const T*>(p.K) + (size_t)b * p.Lk * p.ldk + h * D;
    const T* vb1 = reinterpret_cast<const T*>(p.V) + (size_t)b * p.Lk * p.ldv + h * D;
    const T* ones = reinterpret_cast<const T*>(g_ones_f16);

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

    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const unsigned smem_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const T* d_src[DMA_PER_WAVE];
    int d_ld[DMA_PER_WAVE], d_dst[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int inst = wv + NW * i;
        const int sb = inst / N_DMA1, j = inst - sb * N_DMA1;
        const int o = j * 1024 + lane * 16;
        d_dst[i] = sb * STAGE1 + j * 1024;
        if (j < K_DMA) {
            const int c = o >> 4;
            const int key = c / 5, col = (c - key * 5) * 8;
            d_ld[i] = p.ldk;
            d_src[i] = kb1 + (size_t)(key + sb * KB) * p.ldk + col;
        } else {
            const int c = (o - K_BYTES) >> 4;
            const int r = c / 6, col = c - r * 6;
            const int key = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
            d_ld[i] = col < 5 ? p.ldv : 0;
            d_src[i] = col < 5 ? vb1 + (size_t)(key + sb * KB) * p.ldv + col * 8 : ones;
        }
    }
    auto dma = [&](int stage_off) {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int inst = wv + NW * i;
            if (inst < N_DMA) {
                attn_dma16(d_src[i], smem_base + stage_off + d_dst[i]);
                d_src[i] += (size_t)(KB * SUB) * d_ld[i];
            }
        }
    };
    const int n_tiles = p.Lk / (KB * SUB);         // stages; lab: Lk % 128 == 0
    const int n_mine = (N_DMA - wv + NW - 1) / NW;

    const int k_off01 = l32 * K_ROW + hi * 16;
    const int k_off2 = l32 * K_ROW + 64;
    const int v_off = K_BYTES + (lane >> 2) * V_ROW + (lane & 3) * 8;

    f32x4 oacc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f) oacc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = 0.f;
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    struct S2 { f32x16 a, b; };
    struct PB { unsigned w[2][2][4]; };
    struct KF { V8 a[3], b[3]; };
    struct VF { s16x4 lo[2][3], up[2][3]; };

    auto load_k = [&](const unsigned char* St) -> KF {
        KF k;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int o = st < 2 ? k_off01 + 32 * st : k_off2;
            k.a[st] = *reinterpret_cast<const V8*>(St + o);
            k.b[st] = *reinterpret_cast<const V8*>(St + o + 32 * K_ROW);
        }
        return k;
    };
    auto qk_mfma = [&](const KF& k) -> S2 {
        S2 s;
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s.a) : "v"(k.a[0]), "v"(qf[0]), "v"(negm));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s.b) : "v"(k.b[0]), "v"(qf[0]), "v"(negm));
#pragma unroll
        for (int st = 1; st < 3; ++st) {
            s.a = F16Tag::mfma32(k.a[st], qf[st], s.a);
            s.b = F16Tag::mfma32(k.b[st], qf[st], s.b);
        }
        return s;
    };
    auto load_v = [&](const unsigned char* St) -> VF {
        VF v;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (TRB) {
                typedef __attribute__((address_space(3))) s16x4* lp;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    v.lo[kb][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW + 32 * i));
                    v.up[kb][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW + 32 * i + 16 * V_ROW));
                }
            } else {
                const unsigned va0 = (unsigned)(uintptr_t)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW);
                asm volatile("ds_read_b64_tr_b16 %0, %6\n\t"
                             "ds_read_b64_tr_b16 %1, %6 offset:%7\n\t"
                             "ds_read_b64_tr_b16 %2, %6 offset:32\n\t"
                             "ds_read_b64_tr_b16 %3, %6 offset:%8\n\t"
                             "ds_read_b64_tr_b16 %4, %6 offset:64\n\t"
                             "ds_read_b64_tr_b16 %5, %6 offset:%9\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(v.lo[kb][0]), "=&v"(v.up[kb][0]), "=&v"(v.lo[kb][1]), "=&v"(v.up[kb][1]), "=&v"(v.lo[kb][2]), "=&v"(v.up[kb][2])
                             : "v"(va0), "i"(16 * V_ROW), "i"(16 * V_ROW + 32), "i"(16 * V_ROW + 64)
                             : "memory");
            }
        }
        return v;
    };

    auto softmax_max = [&](S2& s, bool first) {
        float mx;
        if constexpr (TREE) {
            float t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = fmaxf(fmaxf(s.a[2 * r], s.a[2 * r + 1]), fmaxf(s.b[2 * r], s.b[2 * r + 1]));
            mx = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(t[6], t[7])));
        } else {
            mx = fmaxf(s.a[0], s.b[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s.a[r]), s.b[r]);
        }
        mx = mve_max_xor32(mx);
        if (__builtin_expect(first || __any(mx > 0.f), 0)) {
            const float delta = first ? mx : fmaxf(mx, 0.f);
            m_run += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s.a[r] -= delta; s.b[r] -= delta; negm[r] = -m_run; }
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                const auto ar = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
                const float a0 = __uint_as_float(ar[0]), a1 = __uint_as_float(ar[1]);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { oacc[i][0][r] *= a0; oacc[i][1][r] *= a1; }
            }
        }
    };
    auto softmax_exp = [&](const S2& s) -> PB {
        PB pb;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f32x16& sk = kb ? s.b : s.a;
            unsigned pk[8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                constexpr int POLY = (VAR >> 3) & 3;                // 0: none, 1: one pair in 4, 2: one pair in 2, 3: every pair
                const bool poly = POLY == 3 || (POLY == 2 && (r2 & 1)) || (POLY == 1 && (r2 & 3) == 3);
                f32x2 e;
                if (poly) e = exp2_poly_pair(sk[2 * r2], sk[2 * r2 + 1]);
                else e = f32x2{__builtin_amdgcn_exp2f(sk[2 * r2]), __builtin_amdgcn_exp2f(sk[2 * r2 + 1])};
                pk[r2] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, T2));
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ai = v < 2 ? v : v + 2;
                const auto r = __builtin_amdgcn_permlane16_swap(pk[ai], pk[ai + 2], false, false);
                pb.w[kb][0][v] = r[0];
                pb.w[kb][1][v] = r[1];
            }
        }
        return pb;
    };
    auto pv_mfma = [&](const VF& v, const PB& pb) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const V8 va = __builtin_bit_cast(V8, __builtin_shufflevector(v.lo[kb][i], v.up[kb][i], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const u32x4 pw = {pb.w[kb][f][0], pb.w[kb][f][1], pb.w[kb][f][2], pb.w[kb][f][3]};
                    oacc[i][f] = F16Tag::mfma16(va, __builtin_bit_cast(V8, pw), oacc[i][f]);
                }
            }
    };

    auto wait_sync = [&](int keep) {
        if (keep >= 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (keep == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (keep == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto next_stage = [&](int off) { return off + STAGE == NST * STAGE ? 0 : off + STAGE; };

    constexpr int PD = NST - 1;
#pragma unroll
    for (int i = 0; i < PD; ++i)
        if (i < n_tiles) dma(i * STAGE);
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]));

    if constexpr (!PIPE) {
        wait_sync(n_mine * ((n_tiles < PD ? n_tiles : PD) - 1));
        int cur = 0, nxt = PD * STAGE;
        if (nxt == NST * STAGE) nxt = 0;
        for (int t = 0; t < n_tiles; ++t) {
            if (t + PD < n_tiles) dma(nxt);
#pragma nounroll
            for (int sb = 0; sb < SUB; ++sb) {                      // not unrolled: the two tiles' fragments interleaved cost 30 spilled registers
                const unsigned char* St = smem + cur + sb * STAGE1;
                const KF kf = load_k(St);
                S2 s = qk_mfma(kf);
                softmax_max(s, t == 0 && sb == 0);
                const PB pb = softmax_exp(s);
                const VF vf = load_v(St);
                pv_mfma(vf, pb);
            }
            const int last_issued = t + PD < n_tiles ? t + PD : n_tiles - 1;
            wait_sync(last_issued > t + 1 ? n_mine * (last_issued - (t + 1)) : 0);
            nxt = cur;
            cur = next_stage(cur);
        }
    } else {
        {   // tiles 0 and 1 landed (later ones may stay in flight)
            const int issued = n_tiles < PD ? n_tiles : PD;
            wait_sync(issued > 2 ? n_mine * (issued - 2) : 0);
        }
        int cur = 0, nx1 = STAGE, nxt = (PD % NST) * STAGE;
        S2 sA, sB;
        {
            const KF k0 = load_k(smem);
            sA = qk_mfma(k0);
        }
        const int n_main = n_tiles - 1;
        auto step = [&](S2& sc_, S2& sn_, int t) {
            if (t + PD < n_tiles) dma(nxt);
            const KF kn = load_k(smem + nx1);                     // K fragments of tile t + 1 and V^T fragments of tile t: issued up front,
            const VF vf = load_v(smem + cur);                     // consumed behind the maximum / the exponentials
            softmax_max(sc_, t == 0);
            sn_ = qk_mfma(kn);
            const PB pb = softmax_exp(sc_);
            pv_mfma(vf, pb);
            const int last_issued = t + PD < n_tiles ? t + PD : n_tiles - 1;
            wait_sync(last_issued > t + 2 ? n_mine * (last_issued - (t + 2)) : 0);
            nxt = cur; cur = nx1; nx1 = next_stage(nx1);
        };
        int t = 0;
        for (; t + 2 <= n_main; t += 2) { step(sA, sB, t); step(sB, sA, t + 1); }
        if (t < n_main) { step(sA, sB, t); sA = sB; ++t; }
        softmax_max(sA, t == 0);
        const PB pb = softmax_exp(sA);
        const VF vf = load_v(smem + cur);
        pv_mfma(vf, pb);
    }

#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float l = __shfl(oacc[2][f][0], 32 + l16, 64);
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
                    for (int r = 0; r < 4; ++r) o4[r] = (T)(oacc[i][f][r] * inv);
                    *reinterpret_cast<T4*>(orow + dv) = o4;
                }
            }
        }
    }
    {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
        if (lane == 0 && (blockIdx.x & 63) == 5 && p.prof) {
            atomicAdd(&p.prof[0], t1 - prof_t0);
            atomicAdd(&p.prof[1], r1 - prof_r0);
            atomicAdd(&p.prof[2], 1ull);
        }
    }
}


// ---- k_attn6: two 32-query blocks per wave (K / V fragments, LDS-DMA and barriers shared) -----------------------------------------------
template <int NW, int NST, int WPS, int VAR>
__global__ __launch_bounds__(64 * NW, WPS) void k_attn6(const AttnParams p) {
    constexpr int D = 40, KB = 64, QB = 64 * NW;      // two 32-query blocks per wave: K / V fragments, DMA and barriers shared
    constexpr int K_ROW = 80, V_ROW = 96;
    constexpr int K_BYTES = KB * K_ROW, V_BYTES = KB * V_ROW, STAGE = K_BYTES + V_BYTES;
    constexpr int N_DMA = STAGE / 1024, K_DMA = K_BYTES / 1024, DMA_PER_WAVE = (N_DMA + NW - 1) / NW;
    constexpr bool PIPE = (VAR & 1) != 0, TREE = (VAR & 2) != 0, TRB = (VAR & 4) != 0;
    

    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l32 = lane & 31, hi = lane >> 5, l16 = lane & 15, g = lane >> 4;
    unsigned long long prof_t0 = 0, prof_r0 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t0), "=s"(prof_r0) :: "memory");

    const int q_tiles = (p.Lq + QB - 1) / QB;
    const unsigned nblk = (unsigned)(q_tiles * p.heads * p.B);
    const unsigned bid = mve_xcd_remap(blockIdx.x, nblk);
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % p.heads;
    const int b = bid / (q_tiles * p.heads);

    const T* kb1 = reinterpret_cast<const T*>(p.K) + (size_t)b * p.Lk * p.ldk + h * D;
    const T* vb1 = reinterpret_cast<const T*>(p.V) + (size_t)b * p.Lk * p.ldv + h * D;
    const T* ones = reinterpret_cast<const T*>(g_ones_f16);

    const int q_base = qt * QB + wid * 64;
    V8 qf[2][3];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int q = q_base + qb * 32 + l32;
        q = q < p.Lq ? q : p.Lq - 1;
        const T* row = reinterpret_cast<const T*>(p.Q) + ((size_t)b * p.Lq + q) * p.ldq + h * D;
        qf[qb][0] = __builtin_bit_cast(V8, *reinterpret_cast<const u32x4*>(row + 8 * hi));
        qf[qb][1] = __builtin_bit_cast(V8, *reinterpret_cast<const u32x4*>(row + 16 + 8 * hi));
        u32x4 t = {0u, 0u, 0u, 0u};
        if (hi == 0) t = *reinterpret_cast<const u32x4*>(row + 32);
        qf[qb][2] = __builtin_bit_cast(V8, t);
    }

    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const unsigned smem_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const T* d_src[DMA_PER_WAVE];
    int d_ld[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int inst = wv + NW * i;
        const int o = inst * 1024 + lane * 16;
        if (inst < K_DMA) {
            const int c = o >> 4;
            const int key = c / 5, col = (c - key * 5) * 8;
            d_ld[i] = p.ldk;
            d_src[i] = kb1 + (size_t)key * p.ldk + col;
        } else {
            const int c = (o - K_BYTES) >> 4;
            const int r = c / 6, col = c - r * 6;
            const int key = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
            d_ld[i] = col < 5 ? p.ldv : 0;
            d_src[i] = col < 5 ? vb1 + (size_t)key * p.ldv + col * 8 : ones;
        }
    }
    auto dma = [&](int stage_off) {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int inst = wv + NW * i;
            if (inst < N_DMA) {
                attn_dma16(d_src[i], smem_base + stage_off + inst * 1024);
                d_src[i] += (size_t)KB * d_ld[i];
            }
        }
    };
    const int n_tiles = p.Lk / KB;                 // lab: Lk % 64 == 0
    const int n_mine = (N_DMA - wv + NW - 1) / NW;

    const int k_off01 = l32 * K_ROW + hi * 16;
    const int k_off2 = l32 * K_ROW + 64;
    const int v_off = K_BYTES + (lane >> 2) * V_ROW + (lane & 3) * 8;

    f32x4 oacc[2][3][2];
    float m_run[2] = {0.f, 0.f};
    f32x16 negm[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int f = 0; f < 2; ++f) oacc[qb][i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qb][r] = 0.f;
    }

    struct S2 { f32x16 a, b; };
    struct PB { unsigned w[2][2][4]; };
    struct KF { V8 a[3], b[3]; };
    struct VF { s16x4 lo[2][3], up[2][3]; };

    auto load_k = [&](const unsigned char* St) -> KF {
        KF k;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int o = st < 2 ? k_off01 + 32 * st : k_off2;
            k.a[st] = *reinterpret_cast<const V8*>(St + o);
            k.b[st] = *reinterpret_cast<const V8*>(St + o + 32 * K_ROW);
        }
        return k;
    };
    auto qk_mfma = [&](const KF& k, int qb) -> S2 {
        S2 s;
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s.a) : "v"(k.a[0]), "v"(qf[qb][0]), "v"(negm[qb]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s.b) : "v"(k.b[0]), "v"(qf[qb][0]), "v"(negm[qb]));
#pragma unroll
        for (int st = 1; st < 3; ++st) {
            s.a = F16Tag::mfma32(k.a[st], qf[qb][st], s.a);
            s.b = F16Tag::mfma32(k.b[st], qf[qb][st], s.b);
        }
        return s;
    };
    auto load_v = [&](const unsigned char* St) -> VF {
        VF v;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (TRB) {
                typedef __attribute__((address_space(3))) s16x4* lp;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    v.lo[kb][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW + 32 * i));
                    v.up[kb][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW + 32 * i + 16 * V_ROW));
                }
            } else {
                const unsigned va0 = (unsigned)(uintptr_t)(lds_ptr_t)(St + v_off + kb * 32 * V_ROW);
                asm volatile("ds_read_b64_tr_b16 %0, %6\n\t"
                             "ds_read_b64_tr_b16 %1, %6 offset:%7\n\t"
                             "ds_read_b64_tr_b16 %2, %6 offset:32\n\t"
                             "ds_read_b64_tr_b16 %3, %6 offset:%8\n\t"
                             "ds_read_b64_tr_b16 %4, %6 offset:64\n\t"
                             "ds_read_b64_tr_b16 %5, %6 offset:%9\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(v.lo[kb][0]), "=&v"(v.up[kb][0]), "=&v"(v.lo[kb][1]), "=&v"(v.up[kb][1]), "=&v"(v.lo[kb][2]), "=&v"(v.up[kb][2])
                             : "v"(va0), "i"(16 * V_ROW), "i"(16 * V_ROW + 32), "i"(16 * V_ROW + 64)
                             : "memory");
            }
        }
        return v;
    };

    auto softmax_max = [&](S2& s, bool first, int qb) {
        float mx;
        if constexpr (TREE) {
            float t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = fmaxf(fmaxf(s.a[2 * r], s.a[2 * r + 1]), fmaxf(s.b[2 * r], s.b[2 * r + 1]));
            mx = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(t[6], t[7])));
        } else {
            mx = fmaxf(s.a[0], s.b[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s.a[r]), s.b[r]);
        }
        mx = mve_max_xor32(mx);
        if (__builtin_expect(first || __any(mx > 0.f), 0)) {
            const float delta = first ? mx : fmaxf(mx, 0.f);
            m_run[qb] += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s.a[r] -= delta; s.b[r] -= delta; negm[qb][r] = -m_run[qb]; }
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                const auto ar = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
                const float a0 = __uint_as_float(ar[0]), a1 = __uint_as_float(ar[1]);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { oacc[qb][i][0][r] *= a0; oacc[qb][i][1][r] *= a1; }
            }
        }
    };
    auto softmax_exp = [&](const S2& s) -> PB {
        PB pb;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f32x16& sk = kb ? s.b : s.a;
            unsigned pk[8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                const float e0 = __builtin_amdgcn_exp2f(sk[2 * r2]);
                const float e1 = __builtin_amdgcn_exp2f(sk[2 * r2 + 1]);
                pk[r2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{e0, e1}, T2));
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ai = v < 2 ? v : v + 2;
                const auto r = __builtin_amdgcn_permlane16_swap(pk[ai], pk[ai + 2], false, false);
                pb.w[kb][0][v] = r[0];
                pb.w[kb][1][v] = r[1];
            }
        }
        return pb;
    };
    auto pv_mfma = [&](const VF& v, const PB& pb, int qb) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const V8 va = __builtin_bit_cast(V8, __builtin_shufflevector(v.lo[kb][i], v.up[kb][i], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const u32x4 pw = {pb.w[kb][f][0], pb.w[kb][f][1], pb.w[kb][f][2], pb.w[kb][f][3]};
                    oacc[qb][i][f] = F16Tag::mfma16(va, __builtin_bit_cast(V8, pw), oacc[qb][i][f]);
                }
            }
    };

    auto wait_sync = [&](int keep) {
        if (keep >= 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (keep == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (keep == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto next_stage = [&](int off) { return off + STAGE == NST * STAGE ? 0 : off + STAGE; };

    constexpr int PD = NST - 1;
#pragma unroll
    for (int i = 0; i < PD; ++i)
        if (i < n_tiles) dma(i * STAGE);
    asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]));

    {
        wait_sync(n_mine * ((n_tiles < PD ? n_tiles : PD) - 1));
        int cur = 0, nxt = PD * STAGE;
        if (nxt == NST * STAGE) nxt = 0;
        for (int t = 0; t < n_tiles; ++t) {
            if (t + PD < n_tiles) dma(nxt);
            const KF kf = load_k(smem + cur);
            S2 s0 = qk_mfma(kf, 0);
            S2 s1 = qk_mfma(kf, 1);
            const VF vf = load_v(smem + cur);
            softmax_max(s0, t == 0, 0);
            const PB pb0 = softmax_exp(s0);
            pv_mfma(vf, pb0, 0);
            softmax_max(s1, t == 0, 1);
            const PB pb1 = softmax_exp(s1);
            pv_mfma(vf, pb1, 1);
            const int last_issued = t + PD < n_tiles ? t + PD : n_tiles - 1;
            wait_sync(last_issued > t + 1 ? n_mine * (last_issued - (t + 1)) : 0);
            nxt = cur;
            cur = next_stage(cur);
        }
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float l = __shfl(oacc[qb][2][f][0], 32 + l16, 64);
        const float inv = 1.0f / l;
        const int q = q_base + qb * 32 + f * 16 + l16;
        if (q < p.Lq) {
            T* orow = reinterpret_cast<T*>(p.O) + ((size_t)b * p.Lq + q) * p.ldo + h * D;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int dv = i * 16 + g * 4;
                if (dv < D) {
                    T4 o4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o4[r] = (T)(oacc[qb][i][f][r] * inv);
                    *reinterpret_cast<T4*>(orow + dv) = o4;
                }
            }
        }
    }
    {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
        if (lane == 0 && (blockIdx.x & 63) == 5 && p.prof) {
            atomicAdd(&p.prof[0], t1 - prof_t0);
            atomicAdd(&p.prof[1], r1 - prof_r0);
            atomicAdd(&p.prof[2], 1ull);
        }
    }
}



// ---- k_attn5: P V on v_mfma_f32_32x32x16 as well -----------------------------------------------------------------------------------------
// O^T[dv][q] += V^T[dv][key] P^T[key][q] with the SAME 32x32 shape as S^T: the B operand P^T wants lane (q = lane & 31, hi) to hold 8 keys of
// a 16-key step, and that is exactly what the S^T accumulator holds after packing pairs (step s of key block kb = registers 8 s .. 8 s + 7 =
// keys 32 kb + 16 s + 4 hi + {0,1,2,3,8,9,10,11}): no v_permlane16_swap, no cross-lane step for alpha either.  The A operand V^T is read with
// two transpose reads per MFMA whose four rows are the four keys of slots 0-3 / 4-7; key k sits in LDS row (k & ~7) | ((k & 3) << 1) | ((k >> 2) & 1)
// (rows of one read 2 apart: 48 banks, conflict-free for 96-byte rows).  dv 0-31 and 32-63 (32-39 real, 40 = the ones row that accumulates the
// softmax denominator, the rest junk that is never stored): 8 MFMAs and 16 transpose reads per 64-key tile.
// VAR bits: 2 = tree-shaped row maximum; 8 = V^T fragment reads issued before the exponentials
template <int NW, int NST, int WPS, int VAR>
__global__ __launch_bounds__(64 * NW, WPS) void k_attn5(const AttnParams p) {
    constexpr int D = 40, KB = 64, QB = 32 * NW;
    constexpr int K_ROW = 80, V_ROW = 96;
    constexpr int K_BYTES = KB * K_ROW, V_BYTES = KB * V_ROW, STAGE = K_BYTES + V_BYTES;
    constexpr int N_DMA = STAGE / 1024, K_DMA = K_BYTES / 1024, DMA_PER_WAVE = (N_DMA + NW - 1) / NW;
    constexpr bool TREE = (VAR & 2) != 0, VEARLY = (VAR & 8) != 0;

    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE + 64];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l32 = lane & 31, hi = lane >> 5;
    unsigned long long prof_t0 = 0, prof_r0 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t0), "=s"(prof_r0) :: "memory");

    const int q_tiles = (p.Lq + QB - 1) / QB;
    const unsigned nblk = (unsigned)(q_tiles * p.heads * p.B);
    const unsigned bid = mve_xcd_remap(blockIdx.x, nblk);
    const int qt = bid % q_tiles;
    const int h = (bid / q_tiles) % p.heads;
    const int b = bid / (q_tiles * p.heads);

    const T* kb1 = reinterpret_cast<const T*>(p.K) + (size_t)b * p.Lk * p.ldk + h * D;
    const T* vb1 = reinterpret_cast<const T*>(p.V) + (size_t)b * p.Lk * p.ldv + h * D;
    const T* ones = reinterpret_cast<const T*>(g_ones_f16);

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

    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const unsigned smem_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const T* d_src[DMA_PER_WAVE];
    int d_ld[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int inst = wv + NW * i;
        const int o = inst * 1024 + lane * 16;
        if (inst < K_DMA) {
            const int c = o >> 4;
            const int key = c / 5, col = (c - key * 5) * 8;
            d_ld[i] = p.ldk;
            d_src[i] = kb1 + (size_t)key * p.ldk + col;
        } else {
            const int c = (o - K_BYTES) >> 4;
            const int r = c / 6, col = c - r * 6;
            const int key = (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3);          // inverse of the row permutation above
            d_ld[i] = col < 5 ? p.ldv : 0;
            d_src[i] = col < 5 ? vb1 + (size_t)key * p.ldv + col * 8 : ones;
        }
    }
    auto dma = [&](int stage_off) {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int inst = wv + NW * i;
            if (inst < N_DMA) {
                attn_dma16(d_src[i], smem_base + stage_off + inst * 1024);
                d_src[i] += (size_t)KB * d_ld[i];
            }
        }
    };
    const int n_tiles = p.Lk / KB;
    const int n_mine = (N_DMA - wv + NW - 1) / NW;

    const int k_off01 = l32 * K_ROW + hi * 16;
    const int k_off2 = l32 * K_ROW + 64;
    // transpose-read address of this lane for (key block 0, step 0, dv block 0, slots 0-3): row of key 4 hi + ((lane & 15) >> 2), piece lane & 3 of
    // the 16-dv group (lane >> 4) & 1.  Other (kb, s, dvb, slots 4-7): + constant offsets (the row permutation is linear in those bits).
    const int v_key = 4 * hi + ((lane & 15) >> 2);
    const int v_row = (v_key & ~7) | ((v_key & 3) << 1) | ((v_key >> 2) & 1);
    const int v_off = K_BYTES + v_row * V_ROW + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    f32x16 ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[i][r] = 0.f;
    float m_run = 0.f;
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    struct S2 { f32x16 a, b; };
    struct PB { unsigned w[2][2][4]; };          // [key block][step][register]
    struct VF { s16x4 lo[2][2][2], up[2][2][2]; };   // [key block][step][dv block]

    auto qk = [&](const unsigned char* St) -> S2 {
        S2 s;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int o = st < 2 ? k_off01 + 32 * st : k_off2;
            const V8 ka = *reinterpret_cast<const V8*>(St + o);
            const V8 kb_ = *reinterpret_cast<const V8*>(St + o + 32 * K_ROW);
            if (st == 0) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s.a) : "v"(ka), "v"(qf[0]), "v"(negm));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s.b) : "v"(kb_), "v"(qf[0]), "v"(negm));
            } else {
                s.a = F16Tag::mfma32(ka, qf[st], s.a);
                s.b = F16Tag::mfma32(kb_, qf[st], s.b);
            }
        }
        return s;
    };
    typedef __attribute__((address_space(3))) s16x4* lp;
    // key 32 kb + 16 s + 8 second + (4 hi + r) -> row: bits 3.. unchanged, so + (32 kb + 16 s + 8 second) rows
    auto v_addr = [&](const unsigned char* St, int kb, int s, int dvb, int second) {
        return (lp)(lds_ptr_t)(St + v_off + (32 * kb + 16 * s + 8 * second) * V_ROW + dvb * 64);
    };
    auto load_v = [&](const unsigned char* St) -> VF {
        VF v;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dvb = 0; dvb < 2; ++dvb) {
                    v.lo[kb][s][dvb] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(v_addr(St, kb, s, dvb, 0));
                    v.up[kb][s][dvb] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(v_addr(St, kb, s, dvb, 1));
                }
        return v;
    };

    auto softmax_max = [&](S2& s, bool first) {
        float mx;
        if constexpr (TREE) {
            float t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = fmaxf(fmaxf(s.a[2 * r], s.a[2 * r + 1]), fmaxf(s.b[2 * r], s.b[2 * r + 1]));
            mx = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(t[6], t[7])));
        } else {
            mx = fmaxf(s.a[0], s.b[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s.a[r]), s.b[r]);
        }
        mx = mve_max_xor32(mx);
        if (__builtin_expect(first || __any(mx > 0.f), 0)) {
            const float delta = first ? mx : fmaxf(mx, 0.f);
            m_run += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s.a[r] -= delta; s.b[r] -= delta; negm[r] = -m_run; }
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);          // this lane's query is the lane's column of O^T too
#pragma unroll
                for (int r = 0; r < 16; ++r) { ob[0][r] *= alpha; ob[1][r] *= alpha; }
            }
        }
    };
    auto softmax_exp = [&](const S2& s) -> PB {
        PB pb;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f32x16& sk = kb ? s.b : s.a;
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                const float e0 = __builtin_amdgcn_exp2f(sk[2 * r2]);
                const float e1 = __builtin_amdgcn_exp2f(sk[2 * r2 + 1]);
                pb.w[kb][r2 >> 2][r2 & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{e0, e1}, T2));
            }
        }
        return pb;
    };
    auto pv = [&](const VF& v, const PB& pb) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u32x4 pw = {pb.w[kb][s][0], pb.w[kb][s][1], pb.w[kb][s][2], pb.w[kb][s][3]};
#pragma unroll
                for (int dvb = 0; dvb < 2; ++dvb) {
                    const V8 va = __builtin_bit_cast(V8, __builtin_shufflevector(v.lo[kb][s][dvb], v.up[kb][s][dvb], 0, 1, 2, 3, 4, 5, 6, 7));
                    ob[dvb] = F16Tag::mfma32(va, __builtin_bit_cast(V8, pw), ob[dvb]);
                }
            }
    };
    auto wait_sync = [&](int keep) {
        if (keep >= 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (keep == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (keep == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto next_stage = [&](int off) { return off + STAGE == NST * STAGE ? 0 : off + STAGE; };

    constexpr int PD = NST - 1;
#pragma unroll
    for (int i = 0; i < PD; ++i)
        if (i < n_tiles) dma(i * STAGE);
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]));
    wait_sync(n_mine * ((n_tiles < PD ? n_tiles : PD) - 1));
    int cur = 0, nxt = PD * STAGE;
    if (nxt == NST * STAGE) nxt = 0;
    for (int t = 0; t < n_tiles; ++t) {
        if (t + PD < n_tiles) dma(nxt);
        S2 s = qk(smem + cur);
        if constexpr (VEARLY) {
            const VF vf = load_v(smem + cur);
            softmax_max(s, t == 0);
            const PB pb = softmax_exp(s);
            pv(vf, pb);
        } else {
            softmax_max(s, t == 0);
            const PB pb = softmax_exp(s);
            const VF vf = load_v(smem + cur);
            pv(vf, pb);
        }
        const int last_issued = t + PD < n_tiles ? t + PD : n_tiles - 1;
        wait_sync(last_issued > t + 1 ? n_mine * (last_issued - (t + 1)) : 0);
        nxt = cur;
        cur = next_stage(cur);
    }

    {
        // softmax denominator: row dv = 40 of O^T = dv block 1, local row 8 = register 4 of the hi = 0 half
        const float l = __shfl(ob[1][4], l32, 64);
        const float inv = 1.0f / l;
        const int q = q_base + l32;
        if (q < p.Lq) {
            T* orow = reinterpret_cast<T*>(p.O) + ((size_t)b * p.Lq + q) * p.ldo + h * D;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {             // dv block 0: rows 8 r4 + 4 hi .. + 3
                T4 o4;
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = (T)(ob[0][4 * r4 + r] * inv);
                *reinterpret_cast<T4*>(orow + 8 * r4 + 4 * hi) = o4;
            }
            T4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = (T)(ob[1][r] * inv);   // dv block 1: rows 32 + 4 hi .. + 3
            *reinterpret_cast<T4*>(orow + 32 + 4 * hi) = o4;
        }
    }
    {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
        if (lane == 0 && (blockIdx.x & 63) == 5 && p.prof) {
            atomicAdd(&p.prof[0], t1 - prof_t0);
            atomicAdd(&p.prof[1], r1 - prof_r0);
            atomicAdd(&p.prof[2], 1ull);
        }
    }
}

// ---- host --------------------------------------------------------------------------------------------------------------------------------
struct Problem {
    int B, L, heads;
    T *q, *k, *v, *o;
    size_t n;
};

static Problem make_problem(int B, int L, int heads, unsigned seed) {
    Problem pr; pr.B = B; pr.L = L; pr.heads = heads;
    const int C = heads * 40;
    pr.n = (size_t)B * L * C;
    std::vector<T> h(3 * pr.n);
    unsigned s = seed;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f; };
    auto gauss = [&]() { float a = 0; for (int i = 0; i < 4; ++i) a += rnd(); return (a - 2.0f) * 1.732f; };
    const float qs = 1.0f / sqrtf(40.0f) * 1.4426950408889634f;
    for (size_t i = 0; i < pr.n; ++i) h[i] = (T)(gauss() * qs * 1.5f);
    for (size_t i = pr.n; i < 3 * pr.n; ++i) h[i] = (T)gauss();
    CK(hipMalloc(&pr.q, pr.n * 2)); CK(hipMalloc(&pr.k, pr.n * 2)); CK(hipMalloc(&pr.v, pr.n * 2)); CK(hipMalloc(&pr.o, pr.n * 2));
    CK(hipMemcpy(pr.q, h.data(), pr.n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(pr.k, h.data() + pr.n, pr.n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(pr.v, h.data() + 2 * pr.n, pr.n * 2, hipMemcpyHostToDevice));
    return pr;
}

static AttnParams params_of(const Problem& pr, unsigned long long* prof) {
    AttnParams p;
    const int C = pr.heads * 40;
    p.Q = pr.q; p.K = pr.k; p.V = pr.v; p.O = pr.o;
    p.ldq = p.ldk = p.ldv = p.ldo = C;
    p.B = pr.B; p.Lq = pr.L; p.Lk = pr.L; p.heads = pr.heads;
    p.prof = prof;
    return p;
}

template <int KID, int NW, int NST, int WPS, int VAR>
static void run_variant(const char* name, const Problem& small, const float* ref_small, const Problem& big, unsigned long long* d_prof) {
    // correctness on the small problem
    {
        AttnParams p = params_of(small, nullptr);
        CK(hipMemset(small.o, 0, small.n * 2));
        const int qpb = (KID == 6 ? 64 : 32) * NW; const unsigned grid = (unsigned)(((p.Lq + qpb - 1) / qpb) * p.heads * p.B);
        if (KID == 4) k_attn4<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p); else if (KID == 5) k_attn5<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p); else k_attn6<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p);
        CK(hipDeviceSynchronize());
        std::vector<T> h(small.n);
        CK(hipMemcpy(h.data(), small.o, small.n * 2, hipMemcpyDeviceToHost));
        // O is [B*L][heads*40]; ref is [(b, h, q)][40]
        double num = 0, den = 0, mx = 0;
        for (int b = 0; b < small.B; ++b)
            for (int hh = 0; hh < small.heads; ++hh)
                for (int q = 0; q < small.L; ++q)
                    for (int d = 0; d < 40; ++d) {
                        const double r = ref_small[(((size_t)b * small.heads + hh) * small.L + q) * 40 + d];
                        const double o = (double)(float)h[((size_t)b * small.L + q) * small.heads * 40 + hh * 40 + d];
                        num += (o - r) * (o - r); den += r * r; mx = std::max(mx, fabs(o - r));
                    }
        printf("%-34s rel-L2 %.2e max|d| %.2e %s | ", name, sqrt(num / den), mx, sqrt(num / den) < 1e-3 ? "ok  " : "FAIL");
    }
    {
        AttnParams p = params_of(big, d_prof);
        const int qpb = (KID == 6 ? 64 : 32) * NW; const unsigned grid = (unsigned)(((p.Lq + qpb - 1) / qpb) * p.heads * p.B);
        if (KID == 4) k_attn4<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p); else if (KID == 5) k_attn5<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p); else k_attn6<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p);
        CK(hipDeviceSynchronize());
        CK(hipMemset(d_prof, 0, 32));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int it = 10;
        CK(hipEventRecord(e0));
        for (int i = 0; i < it; ++i) { if (KID == 4) k_attn4<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p); else if (KID == 5) k_attn5<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p); else k_attn6<NW, NST, WPS, VAR><<<grid, 64 * NW>>>(p); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        unsigned long long pr[4];
        CK(hipMemcpy(pr, d_prof, 32, hipMemcpyDeviceToHost));
        const double waves = (double)std::max(pr[2], 1ull), ntile = big.L / 64;
        const double fl = 4.0 * big.B * big.heads * (double)big.L * big.L * 40;
        printf("%7.3f ms %6.0f TF | per wave %6.0f cyc/tile, clock %4.2f GHz\n", ms, fl / ms / 1e9, pr[0] / waves / ntile, (double)pr[0] / std::max(pr[1], 1ull) / 10.0);
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2;
    Problem small = make_problem(2, 512, 8, 1234u);
    Problem big = make_problem(64, 4096, 8, 99u);
    float* d_ref; CK(hipMalloc(&d_ref, (size_t)small.B * small.heads * small.L * 40 * 4));
    {
        AttnParams p = params_of(small, nullptr);
        const int total = small.B * small.heads * small.L;
        k_ref<<<(total + 63) / 64, 64>>>(p, d_ref);
        CK(hipDeviceSynchronize());
    }
    std::vector<float> ref((size_t)small.B * small.heads * small.L * 40);
    CK(hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long* d_prof; CK(hipMalloc(&d_prof, 32)); CK(hipMemset(d_prof, 0, 32));
    for (int r = 0; r < rounds; ++r) {
#define RUN(KID, NW, NST, WPS, VAR) run_variant<KID, NW, NST, WPS, VAR>("k" #KID " NW" #NW " NST" #NST " WPS" #WPS " VAR" #VAR, small, ref.data(), big, d_prof)
        if (argc > 2 && !strcmp(argv[2], "poly")) {   // round 5: v_exp_f32 against the packed-fp32 polynomial for 0 / 1/4 / 1/2 / all of the columns
            RUN(4, 8, 2, 4, 4);
            RUN(4, 8, 2, 4, 12);
            RUN(4, 8, 2, 4, 20);
            RUN(4, 8, 2, 4, 28);
            continue;
        }
        if (argc > 2 && !strcmp(argv[2], "sub")) {    // round 5: one barrier per 64 keys against one per 128
            RUN(4, 8, 2, 4, 4);
            RUN(4, 8, 2, 4, 36);                  // (4-wave blocks would issue 6 LDS-DMAs per wave and stage: outside wait_sync's counted range)
            continue;
        }
        RUN(4, 8, 2, 4, 4);
        RUN(6, 4, 2, 2, 4);
        RUN(6, 4, 3, 2, 4);
        RUN(6, 8, 2, 2, 4);
        RUN(6, 4, 2, 2, 6);
        RUN(6, 2, 2, 2, 4);
#undef RUN
    }
    return 0;
}
