#!/usr/bin/env python3
"""Generates tools/probe/probe3.hip: co-issue of MFMA and VALU / LDS instructions on one SIMD of gfx950, measured in
SHADER CYCLES (s_memtime), in constant-clock time (s_memrealtime, 100 MHz) and in wall time (HIP events), so that a
DVFS clock change cannot be mistaken for an issue-rate change.

Every loop body is ONE asm statement (hipcc adds nothing inside it); the loop around it is `#pragma unroll 1`.
Stream mini-language (one token per instruction):
   M<i> v_mfma_f32_32x32x16_f16 into accumulator i (0-3)      m<i> v_mfma_f32_16x16x32_f16 into accumulator i (0-7)
   F v_fma_f32 (3 VGPR sources)   U v_mul_f32 (2 sources)   E v_exp_f32   C v_cvt_pk_f16_f32   X v_max3_f32
   P v_permlane16_swap_b32   V v_mov_b32   L ds_read_b128   T ds_read_b64_tr_b16   N s_nop 0   W s_waitcnt lgkmcnt(0)
   A v_add_f32 with an SGPR source   S v_sub_f32 (2 VGPR sources)
Not part of the product.  Usage: python tools/probe/gen_probe3.py > tools/probe/probe3.hip
"""
import sys

KERNELS = []          # (name, waves_per_simd, [stream role0], [stream role1] or None, mfma_per_iter, note)


def body_asm(stream):
    """asm text + bookkeeping for one loop body"""
    lines = []
    xi = 0
    for tok in stream:
        t = tok[0]
        if t == 'M':
            i = int(tok[1:])
            lines.append(f"v_mfma_f32_32x32x16_f16 %{i}, %[a], %[b], %{i}")
        elif t == 'Z':
            i = int(tok[1:])
            lines.append(f"v_mfma_f32_32x32x16_f16 %{i}, %[a], %[b], %[zc]")
        elif t == 'm':
            i = int(tok[1:])
            lines.append(f"v_mfma_f32_16x16x32_f16 %[q{i}], %[a], %[b], %[q{i}]")
        else:
            x = f"%[x{xi % 16}]"
            x1 = f"%[x{(xi + 5) % 16}]"
            x2 = f"%[x{(xi + 11) % 16}]"
            xi += 1
            if t == 'F': lines.append(f"v_fma_f32 {x}, {x}, %[c1], %[c2]")
            elif t == 'U': lines.append(f"v_mul_f32 {x}, {x}, %[c1]")
            elif t == 'A': lines.append(f"v_add_f32 {x}, %[sc], {x}")
            elif t == 'S': lines.append(f"v_sub_f32 {x}, {x}, %[c2]")
            elif t == 'E': lines.append(f"v_exp_f32 {x}, {x}")
            elif t == 'C': lines.append(f"v_cvt_pk_f16_f32 {x}, {x}, {x1}")
            elif t == 'X': lines.append(f"v_max3_f32 {x}, {x}, {x1}, {x2}")
            elif t == 'P': lines.append(f"v_permlane16_swap_b32 {x}, {x1}")
            elif t == 'V': lines.append(f"v_mov_b32 {x}, {x1}")
            elif t == 'L': lines.append("ds_read_b128 %[y], %[addr]")
            elif t == 'T': lines.append("ds_read_b64_tr_b16 %[z], %[addr]")
            elif t == 'N': lines.append("s_nop 0")
            elif t == 'W': lines.append("s_waitcnt lgkmcnt(0)")
            else: raise ValueError(tok)
    lines.append("s_waitcnt lgkmcnt(0)")
    return "\\n\\t".join(lines)


def uses16(stream):
    return any(t[0] == 'm' for t in stream)


def emit_loop(stream):
    asm = body_asm(stream)
    mixed = uses16(stream) and any(t[0] in 'MZ' for t in stream)
    if mixed:        # attention skeleton: 2 MFMA32 accumulators + 6 MFMA16 accumulators, 8 filler registers (30-operand limit)
        acc_ops = ", ".join(f'"+v"(acc[{i}])' for i in range(2)) + ", " + ", ".join(f'[q{i}] "+v"(q[{i}])' for i in range(6))
        x_ops = ", ".join(f'[x{i}] "+v"(x[{i % 8}])' for i in range(8)) + ", " + ", ".join(f'[x{i}] "+v"(x[{i}])' for i in range(8, 8))
        x_ops = ", ".join(f'[x{i}] "+v"(x[{i}])' for i in range(8))
        asm = asm
        for i in range(8, 16):
            asm = asm.replace(f"%[x{i}]", f"%[x{i - 8}]")
        return (f'#pragma unroll 1\n        for (int it = 0; it < iters; ++it)\n'
                f'            asm volatile("{asm}"\n'
                f'                : {acc_ops}, {x_ops}, [y] "=&v"(y), [z] "=&v"(z)\n'
                f'                : [a] "v"(a), [b] "v"(b), [c1] "v"(c1), [c2] "v"(c2), [addr] "v"(addr), [zc] "v"(zc));\n')
    if uses16(stream):
        acc_ops = ", ".join(f'[q{i}] "+v"(q[{i}])' for i in range(8))
        x_ops = ", ".join(f'[x{i}] "+v"(x[{i}])' for i in range(16))
        return (f'#pragma unroll 1\n        for (int it = 0; it < iters; ++it)\n'
                f'            asm volatile("{asm}"\n'
                f'                : {acc_ops}, {x_ops}, [y] "=&v"(y), [z] "=&v"(z)\n'
                f'                : [a] "v"(a), [b] "v"(b), [c1] "v"(c1), [c2] "v"(c2), [sc] "s"(sc), [addr] "v"(addr));\n')
    acc_ops = ", ".join(f'"+v"(acc[{i}])' for i in range(4))
    x_ops = ", ".join(f'[x{i}] "+v"(x[{i}])' for i in range(16))
    return (f'#pragma unroll 1\n        for (int it = 0; it < iters; ++it)\n'
            f'            asm volatile("{asm}"\n'
            f'                : {acc_ops}, {x_ops}, [y] "=&v"(y), [z] "=&v"(z)\n'
            f'                : [a] "v"(a), [b] "v"(b), [c1] "v"(c1), [c2] "v"(c2), [sc] "s"(sc), [addr] "v"(addr), [zc] "v"(zc));\n')


def emit_kernel(name, wps, s0, s1):
    nthreads = 256 * wps
    use16 = uses16(s0) or (s1 is not None and uses16(s1))
    out = []
    out.append(f'__global__ __launch_bounds__({nthreads}) void k_{name}(unsigned long long* out, float seed, int iters) {{')
    out.append('    PROBE_SETUP')
    if s1 is None:
        out.append('    {\n        ' + emit_loop(s0) + '    }')
    else:
        out.append('    if (role == 0) {\n        ' + emit_loop(s0) + '    } else {\n        ' + emit_loop(s1) + '    }')
    out.append('    PROBE_FINISH')
    out.append('}')
    return "\n".join(out)


def add(name, wps, s0, s1=None, nm=0, note=""):
    KERNELS.append((name, wps, s0, s1, nm, note))


def rep(unit, n):
    r = []
    for _ in range(n):
        r += unit
    return r


def mf32(k, f):      # 8 MFMA32 (4 accumulators round-robin) with k fillers f behind each
    r = []
    for i in range(8):
        r.append(f"M{i % 4}")
        r += [f] * k
    return r


def mf16(k, f):      # 16 MFMA16 (8 accumulators) with k fillers behind each
    r = []
    for i in range(16):
        r.append(f"m{i % 8}")
        r += [f] * k
    return r


# ---- group A: one wave per SIMD, k fillers of one type behind every MFMA ----
for f in "FUECXPLT":
    for k in (0, 2, 4, 6, 8):
        if k == 0 and f != 'F':
            continue
        add(f"a32_{f}{k}", 1, mf32(k, f), nm=8, note=f"1 wave/SIMD: MFMA32 + {k} x {f}")
for f in "FECXPLT":
    for k in (0, 1, 2, 3, 4):
        if k == 0 and f != 'F':
            continue
        add(f"a16_{f}{k}", 1, mf16(k, f), nm=16, note=f"1 wave/SIMD: MFMA16 + {k} x {f}")
# ---- group B: filler-only streams, 1 and 2 waves per SIMD ----
for f in "FUECXPLTV":
    add(f"b1_{f}", 1, [f] * 32, note=f"1 wave/SIMD: 32 x {f} only")
    add(f"b2_{f}", 2, [f] * 32, [f] * 32, note=f"2 waves/SIMD: 32 x {f} only (both)")
# ---- group C: two waves per SIMD, role 0 = MFMA only, role 1 = VALU only ----
add("c_M32_idle", 2, mf32(0, 'F'), ['N'] * 8, nm=8, note="2 waves/SIMD: MFMA32 wave | idle wave")
add("c_M32_M32", 2, mf32(0, 'F'), mf32(0, 'F'), nm=8, note="2 waves/SIMD: MFMA32 wave | MFMA32 wave")
for f in "FEXC":
    add(f"c_M32_{f}", 2, mf32(0, 'F'), [f] * 32, nm=8, note=f"2 waves/SIMD: MFMA32 wave (8/iter) | {f} wave (32/iter)")
    add(f"c_M16_{f}", 2, mf16(0, 'F'), [f] * 32, nm=16, note=f"2 waves/SIMD: MFMA16 wave (16/iter) | {f} wave (32/iter)")
# both waves mixed
for k in (2, 4):
    add(f"c_mix32_F{k}", 2, mf32(k, 'F'), mf32(k, 'F'), nm=8, note=f"2 waves/SIMD: both MFMA32 + {k} F")
    add(f"c_mix32_E{k}", 2, mf32(k, 'E'), mf32(k, 'E'), nm=8, note=f"2 waves/SIMD: both MFMA32 + {k} E")
# ---- group D: the d = 40 attention multiset per 64-key x 32-query tile: 6 MFMA32 + 12 MFMA16-equivalent ...
# modelled on MFMA32 only (12 gaps of 32 cycles = 384 cycles): per gap 1.33 X, 2.67 E, 1.33 C, 0.67 P, 0.5 L, 1 T
tile = []
per_gap = [
    ['X', 'E', 'E', 'C', 'T'], ['X', 'E', 'E', 'E', 'C', 'P', 'T'], ['E', 'E', 'X', 'C', 'L', 'T'],
    ['X', 'E', 'E', 'E', 'C', 'T'], ['X', 'E', 'E', 'C', 'P', 'T'], ['E', 'E', 'E', 'X', 'C', 'L', 'T'],
]
for i in range(8):
    tile.append(f"M{i % 4}")
    tile += per_gap[i % 6]
add("d_attn_1w", 1, tile, nm=8, note="1 wave/SIMD: MFMA32 + attention filler mix (5-7 per gap)")
add("d_attn_2w", 2, tile, tile, nm=8, note="2 waves/SIMD: both MFMA32 + attention filler mix")
tile5 = []
for i in range(8):
    tile5.append(f"M{i % 4}")
    tile5 += ['X', 'E', 'E', 'C', 'T']
add("d_attn5_1w", 1, tile5, nm=8, note="1 wave/SIMD: MFMA32 + {X E E C T} per gap")
tile4 = []
for i in range(8):
    tile4.append(f"M{i % 4}")
    tile4 += ['E', 'E', 'C', 'T']
add("d_attn4_1w", 1, tile4, nm=8, note="1 wave/SIMD: MFMA32 + {E E C T} per gap")
# phase-separated (what an un-orchestrated wave does): 8 MFMA then all fillers
sep = [f"M{i % 4}" for i in range(8)] + rep(['X', 'E', 'E', 'C', 'T'], 8)
add("d_sep_1w", 1, sep, nm=8, note="1 wave/SIMD: 8 MFMA32 THEN 8 x {X E E C T}")
add("d_sep_2w", 2, sep, sep, nm=8, note="2 waves/SIMD: both 8 MFMA32 THEN 8 x {X E E C T}")
# anti-phase pair: role 0 MFMA-then-VALU, role 1 VALU-then-MFMA
sep_r = rep(['X', 'E', 'E', 'C', 'T'], 8) + [f"M{i % 4}" for i in range(8)]
add("d_anti_2w", 2, sep, sep_r, nm=8, note="2 waves/SIMD: role0 MFMA-then-VALU | role1 VALU-then-MFMA (no barrier)")
# ---- group E: dependent accumulator chains ----
add("e_dep32_1", 1, ["M0"] * 8, nm=8, note="1 wave/SIMD: MFMA32 same accumulator back to back")
add("e_dep32_2", 1, ["M0", "M1"] * 4, nm=8, note="1 wave/SIMD: MFMA32 two accumulators alternating")
add("e_dep16_1", 1, ["m0"] * 16, nm=16, note="1 wave/SIMD: MFMA16 same accumulator back to back")
add("e_dep16_2", 1, ["m0", "m1"] * 8, nm=16, note="1 wave/SIMD: MFMA16 two accumulators alternating")

# ---- group G: the attention MFMA skeleton at 1 / 2 / 4 waves per SIMD ----
qk_cd = ["M0", "M1", "M0", "M1", "M0", "M1"]
qk_z = ["Z0", "Z1", "M0", "M1", "M0", "M1"]
pv = [f"m{i}" for i in range(6)] * 2
for w in (1, 2, 4):
    add(f"g_qk_cd_{w}w", w, qk_cd * 2, qk_cd * 2, nm=12, note=f"{w} waves/SIMD: 2 x [M32 x6, two dependent chains, C = D]")
    add(f"g_qk_z_{w}w", w, qk_z * 2, qk_z * 2, nm=12, note=f"{w} waves/SIMD: 2 x [M32 x6, chains start from a separate C]")
    add(f"g_pv_{w}w", w, pv * 2, pv * 2, nm=24, note=f"{w} waves/SIMD: 2 x [M16 x12, six accumulators twice]")
    add(f"g_tile_cd_{w}w", w, qk_cd + pv, qk_cd + pv, nm=18, note=f"{w} waves/SIMD: [M32 x6 C=D] + [M16 x12]")
    add(f"g_tile_z_{w}w", w, qk_z + pv, qk_z + pv, nm=18, note=f"{w} waves/SIMD: [M32 x6 from separate C] + [M16 x12]")
    add(f"g_tile_zc_{w}w", w, qk_z + ['N'] * 3 + ['C'] * 16 + pv, qk_z + ['N'] * 3 + ['C'] * 16 + pv, nm=18, note=f"{w} waves/SIMD: [M32 x6 sep. C] + 16 cvt + [M16 x12]")
    full = qk_z + ['X'] * 16 + ['E'] * 32 + ['C'] * 16 + ['P'] * 8 + pv
    add(f"g_tile_full_{w}w", w, full, full, nm=18, note=f"{w} waves/SIMD: [M32 x6] + 16 X + 32 E + 16 C + 8 P + [M16 x12], phase-separated")

HEADER = r'''// GENERATED by tools/probe/gen_probe3.py -- do not edit.  hipcc --offload-arch=gfx950 -O3 probe3.hip -o probe3
// Co-issue of MFMA and VALU / LDS instructions on one SIMD, in shader cycles (s_memtime), 100 MHz ticks (s_memrealtime) and wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define PROBE_SETUP \
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192]; \
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u; \
    h8 a, b; \
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + 0.01f * i + 0.001f * (threadIdx.x & 63)); b[i] = (_Float16)(0.02f * i - 0.07f + 0.002f * (threadIdx.x & 31)); } \
    f16v acc[4]; f4 q[8]; \
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f; \
    for (int i = 0; i < 8; ++i) q[i] = f4{0.f, 0.f, 0.f, 0.f}; \
    float x[16]; \
    for (int i = 0; i < 16; ++i) x[i] = seed * 0.001f * (i + 1) + 1e-4f * threadIdx.x; \
    const float c1 = 0.999f, c2 = 0.0001f; \
    const float sc = __builtin_amdgcn_readfirstlane(__float_as_int(seed)) * 1e-12f; \
    u4 y = {0u, 0u, 0u, 0u}; u2 z = {0u, 0u}; \
    f16v zc; for (int j = 0; j < 16; ++j) zc[j] = -1.0f - 0.01f * j; \
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x & 63) * 16; \
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) & 1; \
    __syncthreads(); \
    unsigned long long t0, r0, t1, r1; \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");

#define PROBE_FINISH \
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory"); \
    float s = 0; \
    for (int i = 0; i < 16; ++i) s += x[i]; \
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15]; \
    for (int i = 0; i < 8; ++i) s += q[i][0] + q[i][3]; \
    s += (float)(y[0] + y[3] + z[0] + z[1]); \
    if (s == 1.2345e-30f) out[0] = 1; \
    if ((threadIdx.x & 63) == 0) { \
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); \
        out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; \
    }

'''

EXTRA = r'''
// ---- power-limited MFMA peak on RANDOM operands: 16 (or 8) MFMAs per iteration over 4 x 4 different random fragment pairs -----------------
template <int BIG, int WPS>
__global__ __launch_bounds__(256 * WPS) void k_mfma_random(unsigned long long* out, const h8* frags, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frags[(i * 64 + lane) & 1023]; b[i] = frags[512 + ((i * 64 + lane) & 511)]; }
    f4 q[16]; f16v acc[4];
    for (int i = 0; i < 16; ++i) q[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    __syncthreads();
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (BIG) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a[i & 3]), "v"(b[(i >> 1) & 3]));
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(q[i]) : "v"(a[i & 3]), "v"(b[i >> 2]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += q[i][0] + q[i][3];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    if (s == 1.2345e-30f) out[0] = 1;
    if (lane == 0) { const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; }
}

template <int BIG, int WPS>
void run_random(const char* name, unsigned long long* d_out, const h8* d_frags, int iters) {
    const int grid = 256, nth = 256 * WPS, nw = grid * nth / 64;
    k_mfma_random<BIG, WPS><<<grid, nth>>>(d_out, d_frags, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) k_mfma_random<BIG, WPS><<<grid, nth>>>(d_out, d_frags, iters);      // ~10 ms: long enough for the power loop to settle
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    std::vector<unsigned long long> h(2 * nw);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double t = 0, q = 0; for (int w = 0; w < nw; ++w) { t += h[2 * w]; q += h[2 * w + 1]; }
    const double flops = 2.0 * 131072 * iters * (double)nw;      // 8 MFMA32 or 16 MFMA16 = 131072 MAC per iteration per wave
    printf("%-46s %d waves/SIMD: %7.1f cycles/iter/wave, clock %5.3f GHz, wall %7.3f ms = %7.1f TFLOP/s\n", name, WPS, t / nw / iters, t / q / 10.0, ms, flops / ms / 1e9);
}
'''

MAIN = r'''
struct Res { double ticks[2], real[2], wall_ms; };

template <class K>
Res run(K kern, int wps, unsigned long long* d_out, int iters) {
    const int grid = 256, nth = 256 * wps, nw = grid * nth / 64;
    kern<<<grid, nth>>>(d_out, 0.5f, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    kern<<<grid, nth>>>(d_out, 0.5f, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * nw);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    Res r; r.wall_ms = ms;
    for (int role = 0; role < 2; ++role) {
        std::vector<double> t, q;
        for (int w = 0; w < nw; ++w) {
            const int wid = w % (nth / 64);
            if (((wid >> 2) & 1) != role) continue;
            t.push_back((double)h[2 * w]); q.push_back((double)h[2 * w + 1]);
        }
        if (t.empty()) { r.ticks[role] = r.real[role] = 0; continue; }
        std::sort(t.begin(), t.end()); std::sort(q.begin(), q.end());
        r.ticks[role] = t[t.size() / 2]; r.real[role] = q[q.size() / 2];
    }
    return r;
}

int main() {
    unsigned long long* d_out; CK(hipMalloc(&d_out, 2 * 8 * 256 * 8 * 2 * 2));
    {   // power-limited MFMA peak: constant-ish operands (the generated streams below) vs random operands
        std::vector<_Float16> hf(1024 * 8);
        unsigned sd = 12345u;
        auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return ((sd >> 8) & 0xFFFF) / 65536.0f; };
        for (auto& v : hf) { float g = 0; for (int i = 0; i < 4; ++i) g += rnd(); v = (_Float16)((g - 2.0f) * 1.732f); }
        h8* d_frags; CK(hipMalloc(&d_frags, hf.size() * 2)); CK(hipMemcpy(d_frags, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            run_random<0, 1>("MFMA 16x16x32 f16, random normal operands", d_out, d_frags, 4096);
            run_random<0, 2>("MFMA 16x16x32 f16, random normal operands", d_out, d_frags, 4096);
            run_random<1, 1>("MFMA 32x32x16 f16, random normal operands", d_out, d_frags, 4096);
            run_random<1, 2>("MFMA 32x32x16 f16, random normal operands", d_out, d_frags, 4096);
        }
        CK(hipMemset(d_frags, 0, hf.size() * 2));
        run_random<0, 1>("MFMA 16x16x32 f16, ZERO operands", d_out, d_frags, 4096);
        run_random<1, 1>("MFMA 32x32x16 f16, ZERO operands", d_out, d_frags, 4096);
    }
    const int iters = 2048;
    printf("# ticks = s_memtime (shader cycles) per iteration per wave; real = s_memrealtime (10 ns units -> ns) per iteration; clock = ticks / real\n");
    printf("%-14s %-64s %9s %9s %7s | %9s %9s | %8s\n", "kernel", "what", "cyc/it r0", "ns/it r0", "GHz", "cyc/it r1", "ns/it r1", "wall us");
'''


def main():
    w = sys.stdout.write
    w(HEADER)
    for name, wps, s0, s1, nm, note in KERNELS:
        w(emit_kernel(name, wps, s0, s1) + "\n\n")
    w(EXTRA)
    w(MAIN)
    for name, wps, s0, s1, nm, note in KERNELS:
        n0 = len(s0)
        n1 = len(s1) if s1 is not None else 0
        w(f'    {{ Res r = run(k_{name}, {wps}, d_out, iters);\n')
        w(f'      const double c0 = r.ticks[0] / iters, n0 = r.real[0] * 10.0 / iters, c1 = r.ticks[1] / iters, n1 = r.real[1] * 10.0 / iters;\n')
        w(f'      printf("%-14s %-64s %9.1f %9.1f %7.3f | %9.1f %9.1f | %8.1f   [instr/iter %d | %d; mfma/iter %d]\\n", "{name}", "{note}", c0, n0, c0 / n0, c1, n1, r.wall_ms * 1e3, {n0}, {n1}, {nm}); }}\n')
    w('    return 0;\n}\n')


if __name__ == '__main__':
    main()
