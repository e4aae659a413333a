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
//   block = 4 waves x 32 query rows; KV tile = 64 keys, staged once per block in LDS.
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

namespace {

constexpr int QB = 128;      // query rows per block
constexpr int KB = 64;       // keys per tile
constexpr int NT = 256;

struct AttnParams {
    const void* Q; const void* K; const void* V; const void* K2; const void* V2; void* O;
    int ldq, ldk, ldv, ldk2, ldv2, ldo;
    int B, Lq, Lk, Lk2, heads;
    float scale_log2e;   // softmax scale * log2(e)
};

template <int DP> __device__ __forceinline__ int k_swz(int row, int chunk) {
    // K tile: [64 keys][DP] 16-bit, DP*2 bytes per row
    if constexpr (DP == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    else return row * (DP * 2) + ((chunk ^ ((row >> 2) & 3)) << 4);
}
__device__ __forceinline__ int v_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <class Tag, int D>
__global__ __launch_bounds__(NT, (D > 80 ? 1 : 2)) void k_attention(const AttnParams p) {
    constexpr int DP = (D + 31) / 32 * 32;      // contraction length of S^T = K Q^T (zero padded)
    constexpr int KS = DP / 32;                 // MFMA k-steps for S^T
    constexpr int DC = D / 8;                   // valid 16-byte chunks per row
    constexpr int DVF = (D + 15) / 16;          // output (dv) fragments
    constexpr int K_BYTES = KB * DP * 2;
    constexpr int V_BYTES = DVF * 16 * 128;
    constexpr int K_TASKS = KB * DC;            // 16-byte chunk loads per K tile
    constexpr int K_PER_T = (K_TASKS + NT - 1) / NT;
    constexpr int V_TASKS = (KB / 2) * DC;      // (key pair, chunk) tasks per V tile
    constexpr int V_PER_T = (V_TASKS + NT - 1) / NT;
    typedef typename Tag::V8 V8;
    typedef typename Tag::T T;

    __shared__ __attribute__((aligned(16))) unsigned char smem[K_BYTES + V_BYTES];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + K_BYTES;

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
    const T* Kp = reinterpret_cast<const T*>(p.K);
    const T* Vp = reinterpret_cast<const T*>(p.V);
    const T* K2p = reinterpret_cast<const T*>(p.K2);
    const T* V2p = reinterpret_cast<const T*>(p.V2);

    // ---- zero the padded parts of the LDS images once (pad d-chunks of K, pad dv rows of V^T) ----
    for (int i = tid; i < (K_BYTES + V_BYTES) / 16; i += NT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};

    // ---- Q fragments (B operand: n = query, k = d) ------------------------------------------------
    V8 qf[2][KS];
    const int q_base = qt * QB + wid * 32;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
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

    // ---- staging registers --------------------------------------------------------------------------
    u32x4 kreg[K_PER_T];
    u32x4 vreg[V_PER_T][2];

    auto key_row = [&](int j, const T* a, int lda, const T* a2, int lda2) -> const T* {
        // clamp so that padded keys read a valid (finite) row; their scores are masked below
        j = j < Ltot ? j : Ltot - 1;
        if (j < p.Lk) return a + ((size_t)b * p.Lk + j) * lda + h * D;
        return a2 + ((size_t)b * p.Lk2 + (j - p.Lk)) * lda2 + h * D;
    };

    auto load_tile = [&](int t) {
        const int j0 = t * KB;
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i) {
            const int task = tid + i * NT;
            if (task < K_TASKS) {
                const int key = task / DC, c = task - key * DC;
                kreg[i] = *reinterpret_cast<const u32x4*>(key_row(j0 + key, Kp, p.ldk, K2p, p.ldk2) + c * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) {
            const int task = tid + i * NT;
            if (task < V_TASKS) {
                const int pair = task / DC, c = task - pair * DC;
                vreg[i][0] = *reinterpret_cast<const u32x4*>(key_row(j0 + 2 * pair, Vp, p.ldv, V2p, p.ldv2) + c * 8);
                vreg[i][1] = *reinterpret_cast<const u32x4*>(key_row(j0 + 2 * pair + 1, Vp, p.ldv, V2p, p.ldv2) + c * 8);
            }
        }
    };

    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i) {
            const int task = tid + i * NT;
            if (task < K_TASKS) {
                const int key = task / DC, c = task - key * DC;
                *reinterpret_cast<u32x4*>(Ks + k_swz<DP>(key, c)) = kreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) {
            const int task = tid + i * NT;
            if (task < V_TASKS) {
                const int pair = task / DC, c = task - pair * DC;
                // key = 2*pair = 32*kk + 16*hh + 4*gg + jj  ->  position 32*kk + 8*gg + 4*hh + jj in the V^T row
                const int key = 2 * pair;
                const int kk = key >> 5, hh = (key >> 4) & 1, gg = (key >> 2) & 3, jj = key & 3;
                const int pos = 32 * kk + 8 * gg + 4 * hh + jj;          // even; pos+1 holds key+1
                const int chunk = pos >> 3, within = (pos & 7) * 2;       // byte offset inside the 16-byte chunk
                const unsigned short* a = reinterpret_cast<const unsigned short*>(&vreg[i][0]);
                const unsigned short* bb = reinterpret_cast<const unsigned short*>(&vreg[i][1]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int dv = c * 8 + e;
                    *reinterpret_cast<unsigned*>(Vs + v_swz(dv, chunk) + within) = (unsigned)a[e] | ((unsigned)bb[e] << 16);
                }
            }
        }
    };

    // ---- running state ------------------------------------------------------------------------------
    f32x4 oacc[DVF][2];
#pragma unroll
    for (int i = 0; i < DVF; ++i) { oacc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    const int n_tiles = (Ltot + KB - 1) / KB;
    load_tile(0);
    __syncthreads();   // zero-fill complete before the first tile is written

    for (int t = 0; t < n_tiles; ++t) {
        store_tile();
        __syncthreads();
        if (t + 1 < n_tiles) load_tile(t + 1);

        // ---- S^T = K Q^T -------------------------------------------------------------------------
        f32x4 s[4][2];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) { s[kf][0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[kf][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const V8 ka = *reinterpret_cast<const V8*>(Ks + k_swz<DP>(kf * 16 + l16, ks * 4 + g));
                s[kf][0] = Tag::mfma16(ka, qf[0][ks], s[kf][0]);
                s[kf][1] = Tag::mfma16(ka, qf[1][ks], s[kf][1]);
            }
        }
        // mask keys past the end (last tile only)
        if ((t + 1) * KB > Ltot) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * KB + kf * 16 + g * 4 + r;
                    if (key >= Ltot) { s[kf][0][r] = -INFINITY; s[kf][1][r] = -INFINITY; }
                }
        }
        // ---- online softmax (per query = per lane column) ------------------------------------------
        V8 pf[2][2];   // [query fragment][k-step of the PV product]
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float mx = -INFINITY;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kf][f][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[f], mx * p.scale_log2e);
            const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
            m_run[f] = m_new;
            float psum = 0.f;
            float pv[4][4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[kf][f][r] * p.scale_log2e - m_new);
                    pv[kf][r] = e;
                    psum += e;
                }
            l_run[f] = l_run[f] * alpha + psum;
#pragma unroll
            for (int i = 0; i < DVF; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[i][f][r] *= alpha;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                V8 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pk[r] = Tag::from_f32(pv[2 * kk][r]);
                    pk[4 + r] = Tag::from_f32(pv[2 * kk + 1][r]);
                }
                pf[f][kk] = pk;
            }
        }
        // ---- O^T += V^T P^T -----------------------------------------------------------------------
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < DVF; ++i) {
                const V8 va = *reinterpret_cast<const V8*>(Vs + v_swz(i * 16 + l16, kk * 4 + g));
                oacc[i][0] = Tag::mfma16(va, pf[0][kk], oacc[i][0]);
                oacc[i][1] = Tag::mfma16(va, pf[1][kk], oacc[i][1]);
            }
        }
        __syncthreads();   // all waves done with this tile before it is overwritten
    }

    // ---- normalise and store ------------------------------------------------------------------------
    typedef T T4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
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

template <class Tag, int D>
int launch(const AttnParams& p, hipStream_t s) {
    const unsigned grid = (unsigned)(((p.Lq + QB - 1) / QB) * p.heads * p.B);
    k_attention<Tag, D><<<grid, NT, 0, s>>>(p);
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

extern "C" int mve_attention(int dtype, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                             const void* K2, int ldk2, const void* V2, int ldv2, void* O, int ldo, int B, int Lq,
                             int Lk, int Lk2, int heads, int head_dim, float scale, void* stream) {
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
    p.scale_log2e = scale * 1.4426950408889634f;
    if (dtype == MVE_F16) return dispatch_d<F16Tag>(p, head_dim, (hipStream_t)stream);
    if (dtype == MVE_BF16) return dispatch_d<BF16Tag>(p, head_dim, (hipStream_t)stream);
    mve_set_error("attention: unsupported dtype %d", dtype);
    return MVE_ERR_ARG;
}
