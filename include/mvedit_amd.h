/*
 * libmvedit_amd — C ABI of the MI355X (gfx950) engine for MVEdit's
 * denoise -> render -> reconstruct hot path.
 *
 * Conventions
 *   - every pointer named d_* (or documented "device") is a device pointer
 *     owned by the CALLER (PyTorch-ROCm storage in the reference host); the
 *     engine never frees or retains caller memory past the call;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is stream-ordered, no call synchronises the device unless its
 *     comment says so;
 *   - return value: 0 on success, negative mve_status on failure;
 *     mve_last_error() returns a thread-local message for the last failure;
 *   - tensors are dense and contiguous in the layout written next to them.
 *
 * Each entry cites the reference interface it replaces (file:line relative to
 * the Lakonik/MVEdit tree).
 */
#ifndef MVEDIT_AMD_H
#define MVEDIT_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVE_API __attribute__((visibility("default")))

typedef enum {
    MVE_OK = 0,
    MVE_ERR_ARG = -1,     /* bad argument / unsupported shape */
    MVE_ERR_HIP = -2,     /* HIP runtime error (message has the hipError string) */
    MVE_ERR_STATE = -3,   /* object used in the wrong state (e.g. weights missing) */
    MVE_ERR_NOMEM = -4    /* caller-provided workspace / capacity too small */
} mve_status;

typedef enum {
    MVE_F32 = 0,
    MVE_F16 = 1,
    MVE_BF16 = 2,
    MVE_I32 = 3,
    MVE_U8 = 4
} mve_dtype;

MVE_API const char* mve_last_error(void);
MVE_API int mve_version(void);                 /* ABI version, bumped on breaking change (library housekeeping: no reference counterpart) */
MVE_API int mve_device_info(int* n_cu, int* wave_size, char* arch, int arch_len); /* hipGetDeviceProperties of the current device */

/* =========================================================================
 * 1. Ray marching (replaces lib/ops/raymarching: bindings.cpp:5-19, kernels
 *    in src/raymarching.cu).  All floating point is fp32 as in the reference
 *    (raymarching.py:33 `custom_fwd(cast_inputs=torch.float32)`).
 * ========================================================================= */

/* raymarching.cu:92-145 kernel_near_far_from_aabb.
 * rays_o, rays_d: [N,3] f32; aabb: [6] f32 device (xmin,ymin,zmin,xmax,ymax,zmax);
 * nears, fars: [N] f32 out.  Miss -> both FLT_MAX. */
MVE_API int mve_near_far_from_aabb(const float* d_rays_o, const float* d_rays_d, const float* d_aabb,
                                   uint32_t N, float min_near, float* d_nears, float* d_fars, void* stream);

/* raymarching.cu:214-226 / :237-254.  coords: [N,3] i32, indices: [N] i32. Bit exact. */
MVE_API int mve_morton3d(const int32_t* d_coords, uint32_t N, int32_t* d_indices, void* stream);
MVE_API int mve_morton3d_invert(const int32_t* d_indices, uint32_t N, int32_t* d_coords, void* stream);

/* raymarching.cu:268-289 kernel_packbits.  grid: [n_bytes*8] f32, bitfield: [n_bytes] u8;
 * bit i of byte n = (grid[8n+i] >= density_thresh). */
MVE_API int mve_packbits(const float* d_grid, uint32_t n_bytes, float density_thresh, uint8_t* d_bitfield, void* stream);

/* raymarching.cu:303-319 kernel_flatten_rays.  rays: [N,2] i32 (offset,count); res: [M] i32 */
MVE_API int mve_flatten_rays(const int32_t* d_rays, uint32_t N, uint32_t M, int32_t* d_res, void* stream);

/* Training-mode march (raymarching.cu:338-475, two passes in the reference with a host
 * read of the atomic counter between them, raymarching.py:285-296).
 *
 * Here the two passes are separate stream-ordered calls and the atomicAdd is replaced by a
 * deterministic exclusive prefix sum in ray order, so rays[n] = (sum_{m<n} count[m], count[n]).
 * (The reference's offsets depend on atomic arrival order; any sequential execution of its
 * threads in ray order yields exactly these values.)
 *
 *  _count : pass 1.  Writes rays[N,2] i32 and d_total[0] = M (i32).  d_scratch: >= mve_march_scratch_bytes(N) bytes.
 *  _write : pass 2.  Writes xyzs[M,3], dirs[M,3], ts[M,2] for rays whose (offset+count) <= capacity;
 *           samples past `capacity` rows are dropped (never written).
 */
MVE_API size_t mve_march_scratch_bytes(uint32_t N);
MVE_API int mve_march_rays_train_count(const float* d_rays_o, const float* d_rays_d, const uint8_t* d_grid,
                                       float bound, int contract, float dt_gamma, uint32_t max_steps,
                                       uint32_t N, uint32_t C, uint32_t H,
                                       const float* d_nears, const float* d_fars, const float* d_noises,
                                       int32_t* d_rays, int32_t* d_total, void* d_scratch, void* stream);
MVE_API int mve_march_rays_train_write(const float* d_rays_o, const float* d_rays_d, const uint8_t* d_grid,
                                       float bound, int contract, float dt_gamma, uint32_t max_steps,
                                       uint32_t N, uint32_t C, uint32_t H,
                                       const float* d_nears, const float* d_fars, const float* d_noises,
                                       const int32_t* d_rays, uint32_t capacity,
                                       float* d_xyzs, float* d_dirs, float* d_ts, void* stream);

/* raymarching.cu:501-579 / :606-695.  sigmas[M], rgbs[M,3], ts[M,2], rays[N,2];
 * weights[M] must be zero-initialised by the caller (raymarching.py:330). */
MVE_API int mve_composite_rays_train_forward(const float* d_sigmas, const float* d_rgbs, const float* d_ts,
                                             const int32_t* d_rays, uint32_t M, uint32_t N, float T_thresh, int binarize,
                                             float* d_weights, float* d_weights_sum, float* d_depth, float* d_image,
                                             void* stream);
MVE_API int mve_composite_rays_train_backward(const float* d_grad_weights, const float* d_grad_weights_sum,
                                              const float* d_grad_depth, const float* d_grad_image,
                                              const float* d_sigmas, const float* d_rgbs, const float* d_ts,
                                              const int32_t* d_rays, const float* d_weights_sum, const float* d_depth,
                                              const float* d_image, uint32_t M, uint32_t N, float T_thresh, int binarize,
                                              float* d_grad_sigmas, float* d_grad_rgbs, void* stream);

/* Inference march / composite (raymarching.cu:714-829 / :843-925).
 * xyzs, dirs: [n_alive*n_step,3]; ts: [n_alive*n_step,2] -- must be zeroed by the caller
 * (raymarching.py:428-430: a zero ts[.,0] marks "no sample").  composite mutates
 * rays_alive / rays_t / weights_sum / depth / image in place. */
MVE_API int mve_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* d_rays_alive, const float* d_rays_t,
                           const float* d_rays_o, const float* d_rays_d, float bound, int contract, float dt_gamma,
                           uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* d_grid,
                           const float* d_nears, const float* d_fars,
                           float* d_xyzs, float* d_dirs, float* d_ts, const float* d_noises, void* stream);
MVE_API int mve_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int binarize,
                               int32_t* d_rays_alive, float* d_rays_t, const float* d_sigmas, const float* d_rgbs,
                               const float* d_ts, float* d_weights_sum, float* d_depth, float* d_image, void* stream);

/* Stream-ordered compaction of the alive list (replaces the boolean-mask indexing
 * `rays_alive[rays_alive >= 0]`, base_volume_renderer.py:322).  Order preserving.
 * d_n_out[0] = number kept.  d_scratch: >= mve_march_scratch_bytes(n_alive). */
MVE_API int mve_compact_alive(const int32_t* d_rays_alive, uint32_t n_alive, int32_t* d_out, int32_t* d_n_out,
                              void* d_scratch, void* stream);

/* Train branch of VolumeRenderer.forward (lib/models/decoders/base_volume_renderer.py:222-243): keep the samples whose
 * composited weight exceeds `threshold`, order preserving, and re-index rays[N][2] = (offset, count) accordingly.
 * d_pref [M+1] receives the reference's `filt_inds` (exclusive prefix count of kept samples, pref[M] = total kept);
 * d_n_out (device int32) the number kept; outputs are caller-allocated with room for M samples. */
MVE_API size_t mve_cull_scratch_bytes(uint32_t M);
MVE_API int mve_cull_samples(const float* d_weights, uint32_t M, float threshold, const int32_t* d_rays, uint32_t N,
                             const float* d_xyzs, const float* d_dirs, const float* d_ts, float* d_out_xyzs, float* d_out_dirs,
                             float* d_out_ts, int32_t* d_out_rays, int32_t* d_pref, int32_t* d_n_out, void* d_scratch,
                             void* stream);

/* Density-grid refresh, the device part of update_extra_state (base_volume_renderer.py:105-177).
 *   density_grid_points : cell coords [N,3] int32 (NULL = every cell of the grid in meshgrid order) + uniform noise [N,3] in
 *                         [0,1) (NULL = cell centres) -> Morton indices [N] and query points
 *                         (coords - (H-1)/2) * 2*bound/H + noise*2*hw - hw,  hw = bound/H   (:129-135)
 *   density_grid_update : tmp[indices] = min(sigmas, FLT_MAX); grid = where(grid >= 0 & tmp >= 0, max(grid*decay, tmp), grid);
 *                         *d_mean = mean(clamp(grid, 0))  (:166-171).  tmp_grid must be pre-filled with -1 by the caller. */
MVE_API int mve_density_grid_points(const int32_t* d_coords, const float* d_noise, uint32_t N, uint32_t grid_size, float bound,
                                    float* d_xyzs, int32_t* d_indices, void* stream);
MVE_API size_t mve_density_grid_scratch_bytes(uint32_t n_cells);
MVE_API int mve_density_grid_update(float* d_density_grid, float* d_tmp_grid, uint32_t n_cells, const float* d_sigmas,
                                    const int32_t* d_indices, uint32_t N, float decay, float* d_mean_density, void* d_scratch,
                                    void* stream);

/* =========================================================================
 * 2. UNet primitives (activations are NHWC = [B*H*W, C] row-major, 16-bit
 *    storage `dtype` in {MVE_F16, MVE_BF16}, fp32 accumulation everywhere).
 *    They replace the third-party arithmetic (diffusers==0.27.2 on cuDNN /
 *    cuBLAS / SDPA) that the reference drives from
 *    lib/models/architecture/diffusers.py:57-164 (unet_enc/unet_dec) and
 *    lib/models/architecture/ip_adapter/attention_processor.py:198-270.
 * ========================================================================= */

#define MVE_GEMM_GEGLU   1   /* W rows interleaved (value,gate): out[m][i] = v[2i]*gelu(v[2i+1]), out width N/2 */
#define MVE_GEMM_OUT_F32 2   /* out is float32 instead of `dtype` */
#define MVE_CONV_W_CHUNK64 4 /* conv weight is [Cout][Cin/64][3][3][64] (needs C1, C2 multiples of 64) */
#define MVE_GEMM_NO_SPLITK 8 /* never split K even if a workspace is given */
#define MVE_CONV_PAD_BR 32    /* stride-2 conv padded on the bottom / right only: diffusers Downsample2D(padding=0) = F.pad(x, (0,1,0,1)) + conv */
#define MVE_GEMM_RES_AFTER_SCALE 16   /* out = (acc + bias) * out_scale + residual (default: residual is added before the scale) */

/* Every Linear / 1x1 Conv2d of the UNet (diffusers Attention, FeedForward, ResnetBlock2D shortcut, Transformer2DModel proj_in/out,
 * TimestepEmbedding; reached through lib/models/architecture/diffusers.py:69-97, :124-162 and the vendored processors,
 * lib/models/architecture/ip_adapter/attention_processor.py:236-262) with its bias / residual / GEGLU epilogue fused:
 * out[m][n] = out_scale * ( sum_k A[m][k]*W[n][k] + bias[n] + rowvec[m/rows_per_vec][n] + residual[m][n] )
 * A: [M][lda] dtype, W: [N][ldw] dtype (torch Linear / 1x1-conv layout; ldw > K selects a column block),
 * out: [M][ldc]; bias: [N] f32 or NULL; rowvec: [ceil(M/rows_per_vec)][ldrv] f32 or NULL (the per-image time
 * embedding of ResnetBlock2D); residual: [M][ldr] dtype or NULL.  N, K, lda, ldw, ldr multiples of 8. */
MVE_API int mve_gemm(int dtype, const void* d_A, int lda, const void* d_W, int ldw, void* d_out, int ldc,
                     int M, int N, int K, const float* d_bias, const float* d_rowvec, int ldrv, int rows_per_vec,
                     const void* d_residual, int ldr, int flags, float out_scale, void* d_workspace,
                     size_t workspace_bytes, int rows_per_image, void* stream);

/* Tuning knob: GEMM / conv launches whose 256 x 320 tiling yields at least `big_min_blocks` (bits 0..24) blocks use the 256-row tile
 * (bit-identical results, so the choice never affects parity or batch invariance).  0 disables it, a negative value only queries.
 * Bit 27 set: the 256-row tile runs the two-stage main loop (csrc/gemm_big.hip) instead of the ping-pong loop (csrc/gemm_pp.hip);
 * bit 30 set (MVE_GEMM_STRICT_SPLITK=1): a launch that fills the chip un-split emulates the K slices of the slice rule inside one block
 * (bitwise equal to split-K + reducer, i.e. to the same rows in a small batch) instead of accumulating K in one chain (the default since
 * round 4: same value up to fp32 summation order, 3.8 ms per 68 ms step faster); bit 29 set: real split-K + reducer there; the two-blocks-per-CU 256 x 160 three-slot tile for dense
 * GEMMs: by default where the dispatcher asks for the narrow tile (launches too small to fill the chip with 320-wide tiles), bit 26 set:
 * wherever it is eligible, bit 28 set: never; bit 25 set: round 2's LDS ring swizzle (A/B aid).  Returns the WHOLE previous word
 * (threshold and option bits), so `old = tune(x); ...; tune(old)` restores every switch.  The defaults come from the environment
 * variables MVE_GEMM_BIG, MVE_GEMM_PP, MVE_GEMM_PP2 (0 never / 1 wherever eligible / 2 narrow launches only) and MVE_GEMM_STRICT_SPLITK. */
MVE_API int mve_gemm_tune(int big_min_blocks);
/* In-kernel K-slice reduction (round 6): a K-sliced GEMM / conv launch that the 256-row ping-pong tile takes folds its slices inside the launch
 * (the blocks of an output tile write their fp32 partial tiles, meet at a per-tile counter, and each folds its share of the tile's rows over the
 * slices in slice order and runs the fused epilogue) instead of leaving them to a second `k_splitk_reduce` launch; small launches (few images per
 * rank) thereby run on the ping-pong loop instead of the 128-row two-stage kernel.  Same slices, same fold order, same epilogue: bit-identical to
 * partials + reducer.  1 (default, MVE_GEMM_RED) on, 0 off, negative only queries.  Returns the previous value. */
MVE_API int mve_gemm_red_tune(int on);
/* The 128-row GEMM / conv kernel with a four-stage LDS ring (k_gemm_deep, csrc/gemm.hip; round 6): launches of at most `max_blocks` blocks -- the
 * K-sliced GEMMs / convs of the deep UNet levels when a rank holds few images, where a block's K loop is a chain of memory round trips -- keep three K
 * tiles in flight instead of one.  Same tile, MFMA order and epilogue as the two-stage loop: bit-identical.  Default 512 (MVE_GEMM_DEEP); 0 turns it
 * off; negative only queries.  Returns the previous value. */
MVE_API int mve_gemm_deep_tune(int max_blocks);
/* Diagnostics: the number of K slices a GEMM / conv launch of this shape runs with under the current switches -- the slice rule's count (a function
 * of rows per image, N, K only), 1 where the un-split launch fills the chip (or the rule's count again in the strict mode, which emulates the slices
 * inside one block), or the smallest count that fills the chip where the rule would over-fill it.  Host logic only. */
MVE_API int mve_gemm_effective_splitk(int M, int N, int K, int rows_per_image);

/* Development aid for csrc/gemm_pp.hip: with `d_buf` (device, 64 uint64 per launched block) set, fp16 320-wide launches of the
 * ping-pong kernel run an instrumented copy and every wave writes its shader-clock sums {L-section work, wait at the L barrier,
 * M-section work, wait at the M barrier, fragment reads, address prep, 0, 0}; NULL turns it off.  (No reference counterpart.) */
MVE_API int mve_gemm_pp_profile(void* d_buf);

/* (No reference counterpart: scheduling detail of mve_gemm / mve_conv3x3.)  Split-K: at the deep UNet levels one image contributes only a few output tiles while K = 9*1280..9*2560; K is then cut
 * into slices that run concurrently and are summed in a fixed order by a second launch.  The slice count depends on
 * (rows_per_image, N, K) only -- never on the batch -- so results are bit-identical however views are chunked or
 * partitioned across GPUs.  It needs an fp32 scratch of mve_gemm_workspace_bytes(...) bytes (0 = this shape never
 * splits; conv: M = B*Ho*Wo, K = 9*(C1+C2), rows_per_image = Ho*Wo); d_workspace NULL or rows_per_image 0 = no split. */
MVE_API size_t mve_gemm_workspace_bytes(int M, int N, int K, int rows_per_image);

/* 3x3 convolution, padding 1, as an implicit GEMM over NHWC input(s):
 *   input = concat_channels(x1[B,Hs,Ws,C1], x2[B,Hs,Ws,C2]) (C2 = 0: single input), optionally
 *   nearest-upsampled 2x (`upsample`; diffusers Upsample2D) and/or strided (`stride` 1|2; Downsample2D);
 *   W: [Cout][3][3][C1+C2] dtype, or with MVE_CONV_W_CHUNK64 [Cout][(C1+C2)/64][3][3][64] (the K order that
 *   keeps the nine taps of a 64-channel slab adjacent; preferred whenever the channel counts allow it);
 *   out: [B*Ho*Wo][ldc]; epilogue as mve_gemm with rows_per_vec = Ho*Wo. */
MVE_API int mve_conv3x3(int dtype, const void* d_x1, int C1, const void* d_x2, int C2, int B, int Hs, int Ws,
                        int stride, int upsample, const void* d_W, int Cout, void* d_out, int ldc,
                        const float* d_bias, const float* d_rowvec, int ldrv, const void* d_residual, int ldr,
                        int flags, float out_scale, void* d_workspace, size_t workspace_bytes, void* stream);

/* ResnetBlock2D tail in one launch: out = conv3x3(x1; W[:, :9*C1]) + conv1x1([x3 | x4]; W[:, 9*C1:]) + bias + bias2 (+ residual).
 * The 1x1 `conv_shortcut` of a resnet whose input and output widths differ (diffusers ResnetBlock2D, reached through
 * lib/models/architecture/diffusers.py:86-97, :139-156) is appended to the K loop of conv2: W is [Cout][9*C1 + C3 + C4] with the 3x3
 * part in MVE_CONV_W_CHUNK64 order followed by the shortcut matrix; x3 / x4 are the (optionally concatenated) block inputs, NHWC
 * at the output resolution.  C1, C3, C4 multiples of 64.  Split-K sizing: mve_gemm_workspace_bytes(M, Cout, 9*C1 + C3 + C4, H*W). */
MVE_API int mve_conv3x3_shortcut(int dtype, const void* x1, int C1, const void* x3, int C3, const void* x4, int C4, int B, int Hs,
                                 int Ws, const void* W, int Cout, void* out, int ldc, const float* bias, const float* bias2,
                                 const void* residual, int ldr, int flags, float out_scale, void* d_workspace,
                                 size_t workspace_bytes, void* stream);

/* ---- nearest-2x upsample + 3 x 3 conv as four 2 x 2 phase convs (round 4) -------------------------------------------------------------------------
 * Replaces, for Upsample2D (diffusers 0.27.2 resnet/upsampling: F.interpolate(scale_factor=2, mode="nearest") then conv 3x3 pad 1; the UNet's
 * up_blocks.*.upsamplers.0 and the VAE decoder's, reached from lib/models/architecture/diffusers.py:57-164 / lib/pipelines/utils.py decode), the
 * fused form mve_conv3x3(..., upsample = 1): output pixel (2 i + py, 2 j + px) reads source rows {i - 1 + py, i + py} and columns
 * {j - 1 + px, j + px} only, the 3 x 3 taps landing on one source pixel add up, so each output parity (py, px) is a 2 x 2 conv over the SOURCE
 * with K = 4 C: 4 / 9 of the multiply-adds.  Zero padding of the upsampled image = zero padding of the source: exact at the borders.
 *   d_W4 : [phase = 2 py + px][Cout][C / 64][2][2][64], the summed taps from mve_pack_upsample_phase_weights (fp32 sums in ascending (row, column)
 *          order, ONE rounding to the storage type).  That rounding is the only arithmetic difference from the 3 x 3 form: the two agree to the
 *          storage precision of the weights, not bit for bit (tests/test_unet_ops.py pins both the exact identity on the rounded phase weights and
 *          the distance to the 3 x 3 form; end to end: tests/rounding_budget_experiment.py --phase).
 *   d_out: [B][2 Hs][2 Ws][Cout] NHWC, dense; d_out_lo (NULL or the same shape in BYTES): the 8-bit low half (lo8, below) in residual_pair mode; needs Cout % 320 == 0.
 * Needs C % 64 == 0, Cout a multiple of 128, Ws a power of two, B Hs Ws >= 64 (mve_upsample_conv_phases_supported = 1); one launch of the
 * ping-pong kernel (2 x 2 window, grouped output rows) for the four phases where B Hs Ws is a multiple of 256, four launches otherwise; K slices by
 * the conv slice rule with 4 Hs Ws rows per image -- batch independent (workspace: *_workspace_bytes). */
MVE_API int mve_upsample_conv_phases_supported(int C, int Cout, int B, int Hs, int Ws);
/* Where B Hs Ws is a multiple of 256 the four phases run as ONE launch of 4 B Hs Ws rows (the phases share the chip like the tiles of any conv: 64
 * images at the 8 x 8 level = one block per CU in one accumulation chain instead of 4 x (64 tiles x 4 K slices + reducer)); same arithmetic per
 * element up to the K-slice policy.  mve_upsample_conv_phases_tune(0): always four launches (A/B; env MVE_PHASES_ONE_LAUNCH); negative: query;
 * returns the previous value. */
MVE_API int mve_upsample_conv_phases_tune(int one_launch);
MVE_API size_t mve_upsample_conv_phases_workspace_bytes(int C, int Cout, int B, int Hs, int Ws);
MVE_API int mve_pack_upsample_phase_weights(int src_dtype, int dst_dtype, const void* d_w_oihw, int Cout, int C, void* d_W4, void* stream);
MVE_API int mve_upsample_conv_phases(int dtype, const void* d_x, int C, int B, int Hs, int Ws, const void* d_W4, int Cout, void* d_out,
                                     const float* d_bias, int flags, void* d_workspace, size_t workspace_bytes, void* d_out_lo, void* stream);

/* ---- residual stream as an unrounded pair (the executor's `residual_pair` mode, mve_unet_set_residual_mode; round 4, the UNet's DEFAULT since round 5) ----
 * The reference's half-precision modules round the residual stream x + f(x) of ResnetBlock2D / BasicTransformerBlock / Transformer2DModel
 * (diffusers 0.27.2, driven from lib/models/architecture/diffusers.py:57-164) to 16 bits after every block: ~30 % of the end-to-end error against
 * fp32 arithmetic (tests/rounding_budget_experiment.py).  In pair mode a stream tensor is stored as hi = round16(x) -- what every MFMA operand
 * read sees, in the tensor's usual place -- and an 8-BIT low half in a companion tensor of the same shape, one byte per element:
 *   lo8 = E5M2( 2^8 * (x - hi) )   (round to nearest even; OCP E5M2 = torch.float8_e5m2; x ~= hi + 2^-8 * lo8, ~14 mantissa bits for fp16 storage).
 * Round 4 kept the low half in 16 bits; 8 bits lose nothing that the end-to-end error can see (8.6e-4 vs 8.7e-4 from fp32 arithmetic at the
 * benchmark shape, 1.23e-3 without a low half) and halve its HBM traffic.  The *_pair entry points are the plain ones plus the companions:
 *   d_residual_lo : lo8 of the residual ([M][ldr] BYTES, NULL = the residual is `d_residual` alone); needs a residual added before the scale;
 *   d_out_lo      : receives lo8(v - round16(v)) next to d_out = round16(v) ([M][ldc] BYTES, NULL = not wanted); 16-bit non-GEGLU outputs only;
 *   d_x*_lo       : lo8 of the normalised inputs (NULL = the input is the 16-bit tensor alone); statistics and the normalisation use hi + lo.
 * With both companions NULL every *_pair call IS the plain call.  A pair launch that is not K-sliced starts its accumulators from the residual
 * pair -- residual + sum_k a w, then + bias -- on EVERY tile the dispatcher may pick (round 5: the 128-row kernel too), so pair launches are
 * bit-identical across tiles exactly as plain launches are (tests/test_unet_ops.py::test_pair_launches_round_identically_on_every_tile); a
 * K-sliced launch adds the pair in the reducer. */
MVE_API int mve_gemm_pair(int dtype, const void* d_A, int lda, const void* d_W, int ldw, void* d_out, int ldc,
                          int M, int N, int K, const float* d_bias, const float* d_rowvec, int ldrv, int rows_per_vec,
                          const void* d_residual, int ldr, int flags, float out_scale, void* d_workspace,
                          size_t workspace_bytes, int rows_per_image, const void* d_residual_lo, void* d_out_lo, void* stream);

/* mve_gemm_pair (no per-image row vector, no GEGLU, unit scale) followed by LayerNorm of the output rows over the N columns:
 *   d_ln_out[m][:] = LayerNorm(row m of the output as a consumer reads it back: hi + lo8 when d_out_lo is given) * gamma + beta   (eps inside the sqrt).
 * Where the launch runs on the 320-wide pair tile (N = 320, M a multiple of 256, bias, no K slices: the residual-stream GEMMs of the 64 x 64 level
 * from 16 images up) the tile that produces a row normalises it in its epilogue -- the row never returns from HBM for its LayerNorm; every other
 * launch is followed by the LayerNorm kernel.  One row arithmetic for both (csrc/ln_core.h): bit-identical results, so the choice may follow the
 * launch geometry.  BasicTransformerBlock norm1 / norm2 / norm3 behind Transformer2DModel.proj_in / attn1.to_out / attn2.to_out (diffusers 0.27.2 as
 * driven from lib/models/architecture/diffusers.py:69-97).  MVE_GEMM_LN_FUSE=0 / mve_gemm_ln_fuse_tune(0): always the separate kernel. */
MVE_API int mve_gemm_pair_ln(int dtype, const void* d_A, int lda, const void* d_W, int ldw, void* d_out, int ldc, int M, int N, int K,
                             const float* d_bias, const void* d_residual, int ldr, void* d_workspace, size_t workspace_bytes, int rows_per_image,
                             const void* d_residual_lo, void* d_out_lo, void* d_ln_out, int ld_ln, const float* d_ln_gamma, const float* d_ln_beta,
                             float ln_eps, void* stream);
MVE_API int mve_gemm_ln_fuse_tune(int on);

MVE_API int mve_conv3x3_pair(int dtype, const void* x1, int C1, const void* x2, int C2, int B, int Hs, int Ws, int stride,
                             int upsample, const void* W, int Cout, void* out, int ldc, const float* bias, const float* rowvec,
                             int ldrv, const void* residual, int ldr, int flags, float out_scale, void* d_workspace,
                             size_t workspace_bytes, const void* d_residual_lo, void* d_out_lo, void* stream);
MVE_API int mve_conv3x3_shortcut_pair(int dtype, const void* x1, int C1, const void* x3, int C3, const void* x4, int C4, int B, int Hs,
                                      int Ws, const void* W, int Cout, void* out, int ldc, const float* bias, const float* bias2,
                                      int flags, float out_scale, void* d_workspace, size_t workspace_bytes, void* d_out_lo, void* stream);
MVE_API int mve_groupnorm_silu_pair(int dtype, const void* d_x1, int C1, const void* d_x2, int C2, int B, int HW, int G,
                                    float eps, const float* d_gamma, const float* d_beta, int silu, void* d_out,
                                    void* d_workspace, const void* d_x1_lo, const void* d_x2_lo, void* stream);
MVE_API int mve_layernorm_pair(int dtype, const void* d_x, int ldx, void* d_y, int ldy, int M, int C,
                               const float* d_gamma, const float* d_beta, float eps, const void* d_x_lo, void* stream);
/* (hi, lo8) = pair of ((a_hi + a_lo) + alpha * b): the ControlNet residual added to a skip tensor of the stream (diffusers.py:110-121); the low halves
 * are lo8 bytes; an all-zero addend leaves the pair untouched bit for bit */
MVE_API int mve_axpy_pair(int dtype, const void* d_a, const void* d_a_lo, const void* d_b, float alpha, void* d_y, void* d_y_lo, size_t n, void* stream);

/* Scaled-dot-product attention over packed projections (no head permutes):
 *   Q row (b,i) at d_Q + (b*Lq+i)*ldq, head h at column h*head_dim; same for K/V with Lk, O with Lq.
 *   Optional second KV segment (K2,V2,Lk2) is logically concatenated after the first along the key
 *   axis (reference attention: lib/pipelines/zero123plus.py:66-69, lib/models/architecture/diffusers.py:646-673).
 *   Cross-image attention (lib/models/architecture/joint_attn.py:13-17) is B/=n, Lq*=n by the caller.
 *   head_dim in {40, 64, 80, 160}.  Replaces F.scaled_dot_product_attention
 *   (lib/models/architecture/ip_adapter/attention_processor.py:246-248). */
MVE_API int mve_attention(int dtype, const void* d_Q, int ldq, const void* d_K, int ldk, const void* d_V, int ldv,
                          const void* d_K2, int ldk2, const void* d_V2, int ldv2, void* d_O, int ldo,
                          int B, int Lq, int Lk, int Lk2, int heads, int head_dim, float scale, void* stream);
/* Experiment knob (no reference counterpart): 0 = the measured kernel configuration; 1 = for head_dim 40 with a single KV segment, 16
 * query rows per wave, 64-key fills and 3-4 waves per SIMD (the d = 40 case is VALU / exp bound at 2 waves per SIMD).  Results of the
 * two variants agree to rounding, not bitwise (the online-softmax rescale points differ); 2 = for head_dim 80 / 160 a K-tile chunk
 * permutation whose fragment reads are free of LDS bank conflicts (same arithmetic: bit-identical results); 4 = a V^T-tile swizzle
 * that also spreads the transposing stores over the LDS banks (single KV segment, any head_dim; bit-identical results); 6 = 2 + 4;
 * 8..11 = for head_dim 40 the kernel with 32x32x16 Q K^T, LDS-DMA-staged row-major V read through the LDS transpose read and
 * v_permlane16_swap for P (8: 4 waves / 2 LDS stages, 9: 8 waves / 2, 10: 4 waves / 3, 11: 8 waves / 3 stages); other head dims keep 0.
 * Negative: query only.  Returns the previous value (MVE_ERR_ARG for values >= 256 in a release build).  tools/ab_attention.py measures them on one box. */
MVE_API int mve_attention_tune(int variant);
/* Development aid (tools/ab_attention_ablate.py), DEVELOPMENT BUILDS ONLY (MVE_ATTN_LAB=1 python -m mvedit_amd.build): bits 8-19 of
 * mve_attention_tune's argument select a timing-only ablation of the head_dim 40 kernel (results are wrong by construction); those launches add
 * per-wave shader-clock and 100 MHz durations to a device accumulator which this call reads into out4 = {shader cycles, 10 ns ticks, waves, 0}
 * and resets.  A release build carries none of those kernels: this call returns MVE_ERR_STATE and mve_attention_tune rejects the bits. */
MVE_API int mve_attention_profile(unsigned long long* out4);
/* The same attention with Q ALREADY multiplied by softmax_scale * log2(e) (= head_dim^-1/2 * 1.442695...): the logits are in log2 units
 * and no per-logit multiply remains.  This is how the UNet / ControlNet executors call it: they fold the factor into the to_q rows when the
 * weights are packed (fp32 multiply, then the one rounding to 16 bit), which replaces the `scale=` handling of
 * F.scaled_dot_product_attention in the reference's processors (lib/models/architecture/ip_adapter/attention_processor.py:246-248,
 * :348-350, :364-366).  For head_dim 40 and variants 8..11 the running maximum is subtracted by the first MFMA of Q K^T itself. */
MVE_API int mve_attention_prescaled(int dtype, const void* d_Q, int ldq, const void* d_K, int ldk, const void* d_V, int ldv,
                                    const void* d_K2, int ldk2, const void* d_V2, int ldv2, void* d_O, int ldo,
                                    int B, int Lq, int Lk, int Lk2, int heads, int head_dim, void* stream);

/* GroupNorm over NHWC input (optionally the channel-concat of two tensors) with optional fused SiLU:
 *   out[B*HW][C1+C2] = act( (x - mean_g) * rstd_g * gamma + beta ), torch.nn.GroupNorm semantics.
 * gamma/beta: [C1+C2] f32.  d_workspace: >= mve_groupnorm_workspace_bytes(B,HW,C,G) bytes.
 * (ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm, conv_norm_out of diffusers 0.27.2,
 * reached from lib/models/architecture/diffusers.py:86-97,139-162.) */
/* Tensors of at most `fused_max_hw` pixels per image whose (image, lcm(C / G, 8)-channel chunk) slice fits one block's registers take the
 * one-launch path (statistics and normalisation out of registers: one read + one write); the choice depends on (HW, C, G) only, never on
 * the batch.  Default 1024 (MVE_GN_FUSED_MAX_HW); 0 disables; negative only queries.  Returns the previous value. */
MVE_API int mve_groupnorm_tune(int fused_max_hw);
MVE_API size_t mve_groupnorm_workspace_bytes(int B, int HW, int C, int G);
MVE_API int mve_groupnorm_silu(int dtype, const void* d_x1, int C1, const void* d_x2, int C2, int B, int HW, int G,
                               float eps, const float* d_gamma, const float* d_beta, int silu, void* d_out,
                               void* d_workspace, void* stream);

/* LayerNorm over the last axis of x[M][ldx] (C <= 2048, C % 8 == 0); gamma/beta f32 [C]: BasicTransformerBlock.norm1/2/3 of
 * diffusers 0.27.2, reached through the Transformer2DModel calls at lib/models/architecture/diffusers.py:91-96, :128-133, :150-155. */
MVE_API int mve_layernorm(int dtype, const void* d_x, int ldx, void* d_y, int ldy, int M, int C,
                          const float* d_gamma, const float* d_beta, float eps, void* stream);

/* Boundary helpers: the reference seam `unet(sample, t, ...)` is NCHW (adapter3d_mixin.py:117-125). */
MVE_API int mve_nchw_to_nhwc(int dst_dtype, int src_dtype, const void* d_x, int B, int C, int H, int W, int Cpad,
                             void* d_y, void* stream);
MVE_API int mve_nhwc_to_nchw(int dst_dtype, int src_dtype, const void* d_x, int ld, int B, int C, int H, int W,
                             void* d_y, void* stream);
/* diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): out[b] = [cos(t*f_k) | sin(t*f_k)] */
MVE_API int mve_timestep_embedding(int dtype, const float* d_t, int B, int dim, void* d_out, void* stream);
MVE_API int mve_silu(int dtype, const void* d_x, void* d_y, size_t n, void* stream);
MVE_API int mve_axpy(int dtype, const void* d_a, const void* d_b, float alpha, void* d_y, size_t n, void* stream); /* y = a + alpha*b */
/* probs[m][:N] = softmax(scores[m][:N]) (fp32 in, `dtype` out): the softmax of the VAE mid-block attention (diffusers Attention with
 * heads = 1, dim_head = 512, reached through self.vae.decode / .encode, lib/pipelines/mvedit_3d_pipeline.py:1260, :1441), which
 * runs as scores = mve_gemm(Q, K) -> mve_softmax_rows -> mve_gemm(P, V^T). */
MVE_API int mve_softmax_rows(int dtype, const float* d_scores, size_t lds, int M, int N, void* d_probs, size_t ldp, void* stream);
/* classifier-free guidance, adapter3d_mixin.py:130-134: out = g*text + (1-g)*uncond (f32) */
MVE_API int mve_cfg_combine(const float* d_uncond, const float* d_text, float guidance_scale, float* d_out, size_t n,
                            void* stream);
/* x0 = (latents_scaled - sqrt(1-abar_t) * noise_pred) / sqrt(abar_t): `pred_original_sample` of the reference's denoise loop
 * (lib/pipelines/mvedit_3d_pipeline.py:1253-1255), fp32 in / out. */
MVE_API int mve_x0_prediction(const float* d_latents_scaled, const float* d_noise_pred, float sqrt_alpha_bar,
                              float sqrt_one_minus_alpha_bar, size_t n, float* d_x0, void* stream);

/* Tone-mapping look-up table of the pipelines (lib/models/decoders/tonemapping.py:33-54): piecewise-linear interpolation between the
 * `steps` knots (lut_x[k], lut_y[k]) with torch.bucketize(right=True) bucket selection.  inverse = 0: Tonemapping.lut (linear != 0:
 * input_mode='linear', x -> log2(max(x, 1e-6)) first); inverse = 1: Tonemapping.inverse_lut (linear != 0: output_mode='linear',
 * exp2 of the result).  fp32, any shape (n elements); tables are device pointers. */
MVE_API int mve_tonemap_lut(const float* d_x, size_t n, const float* d_lut_x, const float* d_lut_y, int steps, int inverse, int linear,
                            float* d_out, void* stream);
/* Its backward: grad_x = grad_out * d lut / d x (the slope of the selected segment times the derivative of the log2 / exp2 around it) --
 * the reference's lut / inverse_lut are differentiable torch expressions and sit inside the optimisation loops
 * (lib/pipelines/mvedit_3d_pipeline.py:419-420, :438-439, :568-571). */
MVE_API int mve_tonemap_lut_backward(const float* d_x, const float* d_grad_out, size_t n, const float* d_lut_x, const float* d_lut_y,
                                     int steps, int inverse, int linear, float* d_grad_x, void* stream);

/* torchvision.transforms.functional.gaussian_blur on `planes` f32 images [H, W] (reflect padding by ksize / 2, normalised float32 kernel;
 * evaluated separably) as the pipelines use it: target-mask blur (lib/pipelines/mvedit_3d_pipeline.py:473, :671) and `highpass`
 * (lib/pipelines/utils.py:187-188: offset + x - blur(x, 6 round(std) + 1, std), applied to normal patches every iteration, :623-624).
 *   d_base == NULL : out = blur(x)                 d_base != NULL : out = offset + base - blur(x)      (highpass: base = x)
 *   adjoint != 0   : the transposed operator (backward of either form: pass the incoming gradient as x, and as base with offset 0
 *                    for highpass).  tmp: planes * H * W floats, distinct from x and out; out may alias base. */
MVE_API int mve_gaussian_blur(const float* d_x, int planes, int H, int W, int ksize, float sigma, int adjoint, const float* d_base,
                              float offset, float* d_tmp, float* d_out, void* stream);

/* lib/ops/shencoder (src/shencoder.cu:28-337, :359-384; sphere_harmonics.py): real spherical harmonics of inputs [B,3] f32 up to `degree`
 * (1..8) -> outputs [B, degree^2], index l^2 + l + m, Condon-Shortley phase, the un-normalised polynomial form (equal to Y_lm on the unit
 * sphere); dy_dx (nullable) [B, 3, degree^2] = d outputs / d (x, y, z).  Backward: grad_inputs [B,3] = sum_ch grad[b][ch] dy_dx[b][d][ch]
 * (written, not accumulated: the reference's wrapper hands in zeros). */
MVE_API int mve_sh_encode(const float* d_inputs, uint32_t B, int degree, float* d_outputs, float* d_dy_dx, void* stream);
MVE_API int mve_sh_encode_backward(const float* d_grad, const float* d_dy_dx, uint32_t B, int degree, float* d_grad_inputs, void* stream);
/* Shading of a batch of rendered views in one pass (lib/pipelines/mvedit_3d_pipeline.py:1372-1384, same expression at :155-168):
 *   n_cv = (2 n0 - 1, 1 - 2 n1, 1 - 2 n2) from normal_fg;  shading = max(light_v . n_cv, 0) * (1 - ambient) + ambient;
 *   tables given : image = lut(inverse_lut(rgb / max(a, 1e-6)) + log2(max(shading, 1e-6))) * a + bg * (1 - a)
 *   tables NULL  : image = rgb * shading + bg * (1 - a)                                   (`self.tonemapping is None`)
 * rgba [n_views * pix][4], normal_fg [n_views * pix][3], cam_lights [n_views][3], image [n_views * pix][3], all f32. */
MVE_API int mve_shade_views(const float* d_rgba, const float* d_normal_fg, const float* d_cam_lights, uint32_t n_views,
                            uint32_t pix_per_view, float ambient_light, float bg_color, const float* d_lut_x, const float* d_lut_y,
                            int steps, float* d_image, void* stream);
/* Per-point Lambertian shading of the mesh path, forward and backward: the `shading_fun`s handed to MeshRenderer.forward
 * (lib/pipelines/mvedit_3d_pipeline.py:410-440):  shading = max(light_i . normal_i, 0) * (1 - ambient) + ambient;
 *   tables given: out = lut(inverse_lut(albedo) + log2(max(shading, 1e-6)))      tables NULL: out = albedo * shading
 * albedo / normal / lights / out [N,3] f32.  d_grad_out == NULL: forward into d_out.  Otherwise: d_grad_albedo / d_grad_normal [N,3]
 * receive the gradients of <grad_out, out> (d_out may be NULL). */
MVE_API int mve_shade_points(const float* d_albedo, const float* d_normal, const float* d_lights, size_t N, float ambient_light,
                             const float* d_lut_x, const float* d_lut_y, int steps, float* d_out, const float* d_grad_out,
                             float* d_grad_albedo, float* d_grad_normal, void* stream);

/* Image-space loss of one NeRF optimisation iteration and its gradients: lib/pipelines/mvedit_3d_pipeline.py:542-603 (`nerf_optim`,
 * from `out_rgbs = outputs['image']...` to `loss = loss + entropy_loss`) with depth_to_normal (lib/core/utils/geometry_utils.py:119-148),
 * TVLoss(power=1.5) (lib/models/losses/tv_loss.py), L1LossMod (lib/models/losses/pixelwise_loss.py) and Tonemapping.lut / inverse_lut.
 * N = P * ps * ps rays, patch-major ([P, ps, ps] flattened); everything f32, device pointers.
 *   shaded            `not is_init or init_shaded` (:558);  lut_steps = 0 with NULL tables: `self.tonemapping is None`
 *   pixel_loss_weight loss_weight of nerf.pixel_loss (1.2, lib/pipelines/utils.py:231); is_init selects the alpha factor 5 / 1 (:580)
 *   patch_w [P]       cam_weights[target_cam_ids] / cam_weights_mean (`target_w`, constant over a patch, :531-532)
 *   patch_lights [P,3] cam_lights[target_cam_ids] (:533-534)
 *   target_n / target_depth  NULL when `use_normal` / `use_depth` is false
 *   weights [M], ts [M,2]    per-sample outputs of composite_rays_train (entropy term, :596-603); M may be 0 */
typedef struct MveReconLossDesc {
    int32_t P, ps, shaded, is_init, lut_steps;
    float ambient_light, bg_color, normal_bg[3];
    float pixel_loss_weight, normal_reg_weight, depth_weight, entropy_weight, bg_width;
    const float *d_lut_x, *d_lut_y;
    const float *d_image, *d_weights_sum, *d_depth;     /* outputs['image'] [N,3], ['weights_sum'] [N], ['depth'] [N] */
    const float *d_weights, *d_ts;
    uint32_t M;
    const float *d_target_dir, *d_target_rgbs, *d_target_m, *d_target_n, *d_target_depth;   /* [N,3] [N,3] [N] [N,3] [N] */
    const float *d_patch_w, *d_patch_lights;
} MveReconLossDesc;
MVE_API size_t mve_recon_loss_workspace_bytes(int P, int ps, uint32_t M);
/* losses[6] = total (`loss` at :603), pixel_rgb_loss, alphas_loss, normal_reg_loss, depth_loss, entropy_loss; out_rgbs [N,3] and
 * out_normals [N,3] are the tensors the patch (LPIPS) losses of :611-627 consume.  The workspace keeps the normals for the backward. */
MVE_API int mve_recon_loss_forward(const MveReconLossDesc* desc, void* d_ws, size_t ws_bytes, float* d_losses, float* d_out_rgbs,
                                   float* d_out_normals, void* stream);
/* Gradients of g_loss * loss + <g_out_rgbs, out_rgbs> + <g_out_normals, out_normals> (either may be NULL) w.r.t. image [N,3],
 * weights_sum [N], depth [N] and weights [M]: what loss.backward() hands to composite_rays_train's backward.  d_g_loss: one f32 on the
 * device (autograd's incoming gradient, read by the kernels so that the host never waits for it), NULL = 1.  Must follow a forward on
 * the same descriptor and workspace.  Pure gathers: no atomics, bitwise reproducible. */
MVE_API int mve_recon_loss_backward(const MveReconLossDesc* desc, void* d_ws, size_t ws_bytes, const float* d_g_out_rgbs,
                                    const float* d_g_out_normals, const float* d_g_loss, float* d_g_image, float* d_g_weights_sum, float* d_g_depth,
                                    float* d_g_weights, void* stream);

/* The same for one MESH optimisation iteration: lib/pipelines/mvedit_3d_pipeline.py:745-782 (`mesh_optim`, from `out_alphas =
 * render_out['rgba']...` to the regularised sum; the two mesh regularisers of that sum are mve_mesh_reg_*).  N = n * size * size pixels,
 * view-major; rgba [N,4], normal [N,3], depth [N] = render_out['rgba' / 'normal' / 'depth'] (depth only feeds the detached cosine
 * weighting of :751-758), target_m_erode [N] the 5x5-eroded mask (:723-724), view_w [n] = cam_weights / cam_weights_mean. */
typedef struct MveMeshLossDesc {
    int32_t n, size, mesh_is_simplified;
    float normal_bg[3];
    float pixel_loss_weight, normal_reg_weight;
    const float *d_rgba, *d_normal, *d_depth;
    const float *d_target_dir, *d_target_rgbs, *d_target_m_erode, *d_target_m_blur, *d_target_n;   /* [N,3] [N,3] [N] [N] [N,3] or NULL */
    const float *d_view_w;
} MveMeshLossDesc;
MVE_API size_t mve_mesh_loss_workspace_bytes(int n, int size);
/* losses[4] = pixel_rgb_loss + alphas_loss + normal_reg_loss, and the three parts (alpha / normal terms 0 when mesh_is_simplified);
 * out_rgbs [N,3] / out_normals [N,3]: the tensors the patch losses of :784-818 cut their patches from. */
MVE_API int mve_mesh_loss_forward(const MveMeshLossDesc* desc, void* d_ws, size_t ws_bytes, float* d_losses, float* d_out_rgbs,
                                  float* d_out_normals, void* stream);
/* Gradients of g_loss * loss + <g_out_rgbs, out_rgbs> + <g_out_normals, out_normals> w.r.t. rgba [N,4] and normal [N,3] (d_g_loss: one
 * f32 on the device, NULL = 1; the ext gradients may be NULL).  One launch, pure gather. */
MVE_API int mve_mesh_loss_backward(const MveMeshLossDesc* desc, void* d_ws, size_t ws_bytes, const float* d_g_out_rgbs,
                                   const float* d_g_out_normals, const float* d_g_loss, float* d_g_rgba, float* d_g_normal, void* stream);

/* Mesh regularisers of the mesh-optimisation loop and their gradients: laplacian_smooth_loss(verts, faces) and
 * normal_consistency(face_normals, faces) of lib/models/decoders/mesh_renderer/base_mesh_renderer.py:55-101 (with
 * compute_edge_to_face_mapping :20-52 and laplacian_uniform :71-91), called per iteration at lib/pipelines/mvedit_3d_pipeline.py:775-776.
 * verts [V,3] f32, faces [F,3] i32 (indices in [0, V)), face_normals [F,3] f32.  losses[0] = mean_i |(D - A) v|_i over the V vertices
 * (A: distinct neighbours); losses[1] = mean over the unique edges of |1 - clamp(n_t0 . n_t1, -1, 1)| with t0 / t1 the faces listing
 * the edge as (min, max) / (max, min), face 0 where a side is missing (the reference's default).  The workspace carries the per-vertex
 * edge buckets from the forward to the backward. */
MVE_API size_t mve_mesh_reg_workspace_bytes(int V, int F);
MVE_API int mve_mesh_reg_forward(const float* d_verts, int V, const int32_t* d_faces, int F, const float* d_face_normals, void* d_ws,
                                 size_t ws_bytes, float* d_losses, void* stream);
/* g_losses: two f32 on the device (autograd's incoming gradients of the two losses), NULL = (1, 1).  g_verts [V,3] receives the Laplacian
 * term's gradient (normal_consistency does not depend on verts directly), g_face_normals [F,3] the consistency term's. */
MVE_API int mve_mesh_reg_backward(const float* d_verts, int V, const int32_t* d_faces, int F, const float* d_face_normals, void* d_ws,
                                  size_t ws_bytes, const float* d_g_losses, float* d_g_verts, float* d_g_face_normals, void* stream);

/* Mesh.auto_normal (lib/models/decoders/mesh_renderer/mesh_utils.py:359-382, seamless=False; once per mesh-optimisation iteration,
 * lib/pipelines/mvedit_3d_pipeline.py:846-847): face_normals [F,3] = normalize(cross(v1 - v0, v2 - v0)), vn [V,3] = normalize of their
 * per-vertex sum (float atomics, like the reference's scatter_add_).  vn_sum [V,3] (the unnormalised sums) is kept for the backward. */
MVE_API int mve_mesh_normals_forward(const float* d_verts, int V, const int32_t* d_faces, int F, float* d_face_normals, float* d_vn_sum,
                                     float* d_vn, void* stream);
/* g_verts [V,3] = gradient w.r.t. the vertices given g_vn [V,3] and / or g_face_normals [F,3] (either may be NULL); scratch: [V,3] f32. */
MVE_API int mve_mesh_normals_backward(const float* d_verts, int V, const int32_t* d_faces, int F, const float* d_vn_sum, const float* d_g_vn,
                                      const float* d_g_face_normals, float* d_scratch, float* d_g_verts, void* stream);

/* =========================================================================
 * 3. UNet2DCondition executor (native runtime behind the reference's UNet seam).
 *    Replaces `self.unet(sample, t, encoder_hidden_states=..., cross_attention_kwargs=...,
 *    down_block_additional_residuals=..., mid_block_additional_residual=...)`
 *    (lib/pipelines/adapter3d_mixin.py:117-125) and `unet_enc` / `unet_dec`
 *    (lib/models/architecture/diffusers.py:57-99 / :102-164).
 *
 *    Topology arguments are diffusers' UNet2DConditionModel config values: per level i,
 *    block_out_channels[i], down_attn[i] (CrossAttnDownBlock2D vs DownBlock2D), num_heads[i],
 *    transformer_layers[i].  Weights are engine-owned: mve_unet_load_param copies one tensor of
 *    the diffusers state dict (by its diffusers name, in its torch layout, any float dtype) into
 *    packed device storage; the caller's tensor is not retained.
 * ========================================================================= */
MVE_API int mve_unet_create(void** handle, int dtype, int in_channels, int out_channels, int n_levels,
                            const int* block_out_channels, int layers_per_block, const int* down_attn,
                            const int* num_heads, const int* transformer_layers, int cross_attention_dim,
                            int norm_num_groups, float norm_eps, int use_linear_projection);
MVE_API int mve_unet_destroy(void* handle);
MVE_API size_t mve_unet_weight_bytes(void* handle);
MVE_API int mve_unet_load_param(void* handle, const char* name, const void* d_src, int src_dtype, int ndim,
                                const long long* shape, void* stream);
/* returns the number of diffusers parameters not loaded yet; buf receives the first missing name */
MVE_API int mve_unet_missing_params(void* handle, char* buf, int buf_len);

/* (No reference counterpart: executor introspection.)  Builds (and caches) the static op list for this problem size.
 * flops[5] = analytic 2*MAC counts per class {conv3x3, linear/1x1, attention, norm, other}; SURVEY.md section 8(d) quotes their sum. */
MVE_API int mve_unet_plan(void* handle, int B, int H, int W, int ctx_len, int num_cross_attn_imgs, int has_residuals,
                          int io_dtype, int residuals_nhwc, size_t* workspace_bytes, int* n_ops, double* flops);

/* phase 0: full forward; 1: unet_enc only (state stays in the workspace); 2: unet_dec only (same workspace).
 * sample: [B, in_channels, H, W] NCHW io_dtype; timesteps: [B] f32 device; ctx: [B, ctx_len, cross_dim] io_dtype;
 * num_cross_attn_imgs: CrossImageAttnProcWrapper group size (lib/models/architecture/joint_attn.py:11-37), 1 = off;
 * down_residuals: host array of n_levels*(layers_per_block+1) device pointers (ControlNet down_block_res_samples, NCHW io_dtype,
 * or NHWC engine dtype if residuals_nhwc) or NULL; out: [B, out_channels, H, W] NCHW io_dtype.
 * op_ms: optional HOST array [n_ops]; when given every op is bracketed with HIP events on `stream` and the call
 * synchronises the stream before returning (profiling mode). */
MVE_API int mve_unet_forward(void* handle, int phase, const void* d_sample, int io_dtype, const float* d_timesteps,
                             const void* d_ctx, int B, int H, int W, int ctx_len, int num_cross_attn_imgs,
                             const void* const* down_residuals, const void* d_mid_residual, int residuals_nhwc,
                             void* d_out, void* d_workspace, size_t workspace_bytes, float* op_ms, void* stream);
/* ControlNetModel (diffusers; lib/pipelines/adapter3d_mixin.py:101-116, :173-186, :279-287 call it through MultiControlNetModel):
 * the UNet's conv_in + down blocks + mid block, a conditioning embedding on the 8H x 8W control image that is added to
 * conv_in(sample), and one 1x1 "zero" convolution per skip / mid tensor.  Same topology arguments and parameter-loading
 * protocol as the UNet (mve_unet_load_param / _missing_params / _plan / _weight_bytes / _destroy work on the handle; parameter
 * names are the ControlNetModel state-dict names).
 *   d_cond       [B, conditioning_channels, 8H, 8W] NCHW in io_dtype ([B / R, ...] after mve_controlnet_set_cond_repeat(handle, R))
 *   d_outputs    n_levels*(layers_per_block+1) + 1 device pointers: down_block_res_samples then mid_block_res_sample, each NHWC
 *                [B*h*w, C] in the ENGINE dtype -- exactly what mve_unet_forward accepts with residuals_nhwc = 1
 *   out_k = conditioning_scale * zero_conv_k(feature_k), or, with accumulate != 0, out_k += that (MultiControlNetModel's sum) */
MVE_API int mve_controlnet_create(void** handle, int dtype, int in_channels, int conditioning_channels, int n_levels,
                                  const int* block_out_channels, int layers_per_block, const int* down_attn, const int* num_heads,
                                  const int* transformer_layers, int cross_attention_dim, int norm_num_groups, float norm_eps,
                                  int use_linear_projection);
/* Shared conditioning images: with repeat = R >= 1, d_cond of mve_controlnet_forward holds B / R images and batch item b uses image b mod (B / R).
 * Under classifier-free guidance both halves of the batch carry the same control images (lib/pipelines/mvedit_3d_pipeline.py:1232, :1417:
 * `ctrl_images.split(diff_bs) * 2`): the conditioning embedding (8 convs on the 8H x 8W image, ~10 % of a ControlNet forward) then runs once for
 * both.  Bit-identical to passing the images R times.  Part of the plan key; returns the previous value (repeat < 1: query).  Not combined with the
 * residual-pair mode. */
MVE_API int mve_controlnet_set_cond_repeat(void* handle, int repeat);
MVE_API int mve_controlnet_forward(void* handle, const void* d_sample, int io_dtype, const float* d_timesteps, const void* d_ctx,
                                   const void* d_cond, int B, int H, int W, int ctx_len, float conditioning_scale, int accumulate,
                                   void* const* d_outputs, void* d_workspace, size_t workspace_bytes,
                                   float* op_ms /* optional host array [n_ops]: per-op milliseconds (synchronises) */, void* stream);

/* AutoencoderKL halves (diffusers==0.27.2 autoencoders/vae.py Decoder / Encoder with post_quant_conv / quant_conv), replacing
 *   self.vae.decode(x0 / scaling_factor, return_dict=False)[0]   lib/pipelines/mvedit_3d_pipeline.py:1258-1262, adapter3d_mixin.py:327-338
 *   self.vae.encode(images * 2 - 1, ...)                          lib/pipelines/mvedit_3d_pipeline.py:1118-1120, :1439-1443
 * half 1: [B, in_channels = latent, H, W] -> [B, out_channels, H * 2^(n-1), W * 2^(n-1)];
 * half 2: [B, in_channels = image, H, W] -> [B, out_channels = 2 * latent, H / 2^(n-1), W / 2^(n-1)] = cat(mean, logvar) (the
 *         DiagonalGaussianDistribution arithmetic on those moments stays with the caller).
 * The handle is an executor handle: parameters arrive through mve_unet_load_param under their AutoencoderKL state-dict names
 * (`decoder.*` + `post_quant_conv.*` for half 1, `encoder.*` + `quant_conv.*` for half 2; a name of the other half is an error),
 * mve_unet_missing_params / mve_unet_op_info / mve_unet_destroy apply unchanged.  ResnetBlock2D without time embedding, eps as
 * given (1e-6 in diffusers); mid-block Attention (1 head of width C) = GEMM q k^T (fp32 scores) -> mve_softmax_rows -> GEMM P V;
 * Downsample2D(padding=0) = MVE_CONV_PAD_BR.  in/out channels <= 8.  d_in / d_out are NCHW in io_dtype (f32 | f16 | bf16).
 * One call handles B images whose widest full-resolution activation has < 2^31 elements (8 images at 512 x 512 for the SD VAE). */
MVE_API int mve_vae_create(void** handle, int dtype, int half, int in_channels, int out_channels, int n_levels,
                           const int* block_out_channels, int layers_per_block, int norm_num_groups, float norm_eps);
MVE_API int mve_vae_plan(void* handle, int B, int H, int W, int io_dtype, size_t* workspace_bytes, int* n_ops, double* flops);
MVE_API int mve_vae_forward(void* handle, const void* d_in, int io_dtype, int B, int H, int W, void* d_out, void* d_workspace,
                            size_t workspace_bytes, float* op_ms, void* stream);

/* SRVGGNetCompact (`image_enhancer` of the pipelines: lib/models/decoders/image_space_ss.py:8-70, built at lib/pipelines/utils.py:212-215
 * with num_feat 64, num_conv 32, upscale 4, PReLU; called on every batch of rendered views below 512 x 512,
 * lib/pipelines/mvedit_3d_pipeline.py:1399-1400).  [B, C, H, W] -> [B, C, H * r, W * r]: conv3x3 + per-channel PReLU stack at the
 * input resolution, conv to C * r * r channels, PixelShuffle(r), plus the nearest-upsampled input.  An executor handle: parameters
 * through mve_unet_load_param under the module's own state-dict names (`body.<i>.weight|bias`), mve_unet_missing_params /
 * mve_unet_op_info / mve_unet_destroy apply.  num_out_ch == num_in_ch <= 8, num_feat % 8 == 0. */
MVE_API int mve_srvgg_create(void** handle, int dtype, int num_in_ch, int num_out_ch, int num_feat, int num_conv, int upscale);
MVE_API int mve_srvgg_plan(void* handle, int B, int H, int W, int io_dtype, size_t* workspace_bytes, int* n_ops, double* flops);
MVE_API int mve_srvgg_forward(void* handle, const void* d_in, int io_dtype, int B, int H, int W, void* d_out, void* d_workspace,
                              size_t workspace_bytes, float* op_ms, void* stream);
/* LPIPS(net='vgg') perceptual loss, forward and backward w.r.t. the prediction: the `patch_loss` of the reference's reconstruct step
 * (lib/models/losses/lpips_loss.py:8-42 -> lpips==0.1.4 `LPIPS.forward`; lib/models/autoencoders/base_nerf.py:337-344, 8 patches of
 * 128 x 128 per optimisation iteration).  An executor handle: parameters through mve_unet_load_param under lpips' own state-dict
 * names (`net.slice<k>.<idx>.weight|bias` = torchvision VGG16 features, `lin<k>.model.1.weight` (or `lins.<k>...`),
 * `scaling_layer.shift|scale`).  normalize_inputs: pred / target arrive in [0, 1] and are mapped to [-1, 1] first (LPIPSLoss default).
 *   forward : loss[n] = sum_l mean_hw sum_c w_lc (u_pred - u_target)^2, u = f / (|f|_c + 1e-10), f = relu{1_2,2_2,3_3,4_3,5_3};
 *             pred, target NCHW [B,3,H,W] in io_dtype, H and W multiples of 16; d_loss f32 [B].
 *   backward: d_grad_pred = d(sum_n grad_loss[n] loss[n]) / d pred, NCHW io_dtype; reads the activations the forward call left in
 *             d_workspace (same workspace, same B/H/W, nothing else run on it in between).  Every VGG conv's data gradient is the
 *             forward implicit-GEMM kernel on a transposed + flipped packing of the same weight. */
MVE_API int mve_lpips_create(void** handle, int dtype, int normalize_inputs);
MVE_API int mve_lpips_plan(void* handle, int B, int H, int W, int io_dtype, size_t* workspace_bytes, int* n_ops, int* n_forward_ops,
                           double* flops);
MVE_API int mve_lpips_forward(void* handle, const void* d_pred, const void* d_target, int io_dtype, int B, int H, int W, float* d_loss,
                              void* d_workspace, size_t workspace_bytes, void* stream);
MVE_API int mve_lpips_backward(void* handle, const float* d_grad_loss, int io_dtype, int B, int H, int W, void* d_grad_pred,
                               void* d_workspace, size_t workspace_bytes, void* stream);
/* LPIPS building blocks (csrc/lpips.hip), NHWC `dtype` activations:
 *   scale        : out[2B*H*W][8] = ((normalize ? 2x-1 : x) - shift_c) / scale_c of pred (first B images) and target, channels 3..7 zero
 *   input_grad   : d x[b][c][y][x] = g[(b,y,x)][c] / scale_c * (normalize ? 2 : 1), NCHW io_dtype
 *   maxpool2x2   : nn.MaxPool2d(2, 2) and its backward (gradient to the first arg-max of the window, as torch)
 *   relu_backward: grad = out > 0 ? grad : 0, in place
 *   lpips_layer  : loss[n] (+)= mean_hw sum_c w_c (u0 - u1)^2 over feat = [pred half | target half]; _backward: d / d feat(pred half) */
MVE_API int mve_lpips_scale(int dtype, int io_dtype, const void* d_pred, const void* d_target, int B, int H, int W, const float* d_shift3,
                            const float* d_scale3, int normalize, void* d_out, void* stream);
MVE_API int mve_lpips_input_grad(int dtype, int io_dtype, const void* d_g8, int B, int H, int W, const float* d_scale3, int normalize,
                                 void* d_out, void* stream);
MVE_API int mve_maxpool2x2(int dtype, const void* d_x, int B, int H, int W, int C, void* d_y, void* stream);
MVE_API int mve_maxpool2x2_backward(int dtype, const void* d_x, const void* d_grad_y, int B, int H, int W, int C, void* d_grad_x,
                                    void* stream);
MVE_API int mve_relu_backward(int dtype, void* d_grad, const void* d_out, size_t n, void* stream);
MVE_API size_t mve_lpips_layer_scratch_bytes(int B, int HW);
MVE_API int mve_lpips_layer(int dtype, const void* d_feat, const float* d_lin_w, int B, int HW, int C, int accumulate, float* d_loss,
                            void* d_scratch, void* stream);
MVE_API int mve_lpips_layer_backward(int dtype, const void* d_feat, const float* d_lin_w, const float* d_grad_loss, int B, int HW, int C,
                                     void* d_grad_feat, void* stream);
/* y = x >= 0 ? x : slope[c] * x over NHWC rows (nn.PReLU(num_parameters=C), image_space_ss.py:41-56); n = rows * C elements */
MVE_API int mve_prelu(int dtype, const void* d_x, const float* d_slope, int C, void* d_y, size_t n, void* stream);
/* out[b][c][y*r+i][x*r+j] = src[(b,y,x)][c*r*r + i*r + j] + base[b][c][y][x]: nn.PixelShuffle(r) of an NHWC fp32 tensor (row stride ld)
 * plus F.interpolate(base, scale_factor=r, mode='nearest') (image_space_ss.py:66-69); base and out NCHW in io_dtype. */
MVE_API int mve_pixel_shuffle_add(int io_dtype, const float* d_src, int ld, const void* d_base, int B, int C, int H, int W, int r,
                                  void* d_out, void* stream);

/* Tuning knob for engines created AFTER the call: 1 (default) folds every ResnetBlock2D conv_shortcut into conv2's K loop
 * (mve_conv3x3_shortcut), 0 keeps the separate 1x1 GEMMs; negative only queries.  Returns the previous setting. */
MVE_API int mve_unet_tune(int fuse_shortcut);
/* Opt-in hipGraph replay of mve_unet_forward (no reference counterpart; the reference launches every kernel eagerly): a forward whose
 * plan and every pointer argument (sample, timesteps, context, output, workspace, residuals, stream) equal those of an earlier call is
 * captured into a graph on its second sighting and replayed afterwards -- for launch-bound small-batch forwards (8 images per rank at
 * 8 GPUs).  Calls with op_ms (profiling) always run eagerly.  enable < 0 only queries; disabling frees the cached graphs.  Returns
 * the previous setting.  Off by default. */
MVE_API int mve_unet_graph(void* handle, int enable);

/* Attention-processor options of the reference, applied to every later plan/forward of this engine:
 *   ip_tokens > 0 : IPAttnProcessor2_0 (lib/models/architecture/ip_adapter/attention_processor.py:301-396) -- the last ip_tokens rows of
 *                   encoder_hidden_states are projected with `<block>.attn2.processor.to_k_ip/to_v_ip.weight` (loaded through
 *                   mve_unet_load_param under those names) and attended to in a second softmax, added with ip_scale.
 *   ref_mode      : ReferenceAttnProc (lib/models/architecture/diffusers.py:646-673) / ReferenceOnlyAttnProc
 *                   (lib/pipelines/zero123plus.py:43-77).  1 = mode 'w': every self-attention layer stores its keys/values;
 *                   2 = modes 'r' / 'm': every self-attention layer appends the stored tokens to its keys/values.  The first
 *                   ref_skip batch items neither store nor read (is_cfg_guidance).  ref_H x ref_W is the latent size of the
 *                   pass that wrote the store (mode 2 only).  d_ref_store is caller-owned device memory of at least
 *                   mve_unet_ref_store_bytes(B, ref_H, ref_W, ref_skip) bytes that must stay valid between the two passes. */
MVE_API int mve_unet_set_attention(void* handle, int ip_tokens, float ip_scale, int ref_mode, int ref_H, int ref_W, int ref_skip,
                                   void* d_ref_store, size_t ref_store_bytes);
/* Residual stream of the UNet / ControlNet executor (no reference counterpart: the reference's half modules round it after every block).
 * pair = 1: x + f(x) of ResnetBlock2D / BasicTransformerBlock / Transformer2DModel is carried as an unrounded (hi, lo) pair of 16-bit tensors -- the
 * residual adds, GroupNorm and LayerNorm read and write the pair, every MFMA operand reads `hi` -- which brings the end-to-end error of one SD-1.5
 * forward at 64 x 64 against fp32 arithmetic from 1.23e-3 to below north_star's 1e-3 (tests/rounding_budget_experiment.py predicts 7.0e-4) for
 * 4 more bytes per stream element and pass.  pair = 0 (default): the single 16-bit tensors of rounds 1-3.  Negative: query.  Returns the previous
 * mode; the mode is part of the plan key. */
MVE_API int mve_unet_set_residual_mode(void* handle, int pair);
MVE_API size_t mve_unet_ref_store_bytes(void* handle, int B, int ref_H, int ref_W, int ref_skip);

/* (No reference counterpart: executor introspection, the source of bench.py's per-kernel roofline figures.)
 * op i of the cached plan: class, flops, label; returns 1 if the op belongs to unet_enc, 2 for unet_dec */
MVE_API int mve_unet_op_info(void* handle, int i, int* cls, double* flops, char* label, int label_len);

/* =========================================================================
 * 4. NeRF render path (float32 throughout, as in the reference).
 * ========================================================================= */

/* iNGPDecoder.point_decode (lib/models/decoders/ingp_decoder.py:106-120): tinycudann HashGrid (Smoothstep, 2 features
 * per level, log2_hashmap_size 19) -> MLP(2L -> hidden -> 4, ReLU) -> sigma = exp(h0 + blob(x)), rgb = sigmoid(h1..3)*(1+2s)-s.
 * level_* are HOST arrays [n_levels] (scale, resolution, row offset, row count per level; see oracle/nerf_oracle.py:grid_meta
 * for tiny-cuda-nn's formulas); d_table: [rows][2] f32; d_w1: [hidden][2L], d_w2: [4][hidden] (torch Linear layout).
 * d_rgbs may be NULL (point_density_decode). n_levels in {12, 14, 16}. */
MVE_API int mve_hashgrid_mlp_decode(const float* d_xyz, uint32_t M, const float* d_table, int n_levels,
                                    const float* level_scale, const uint32_t* level_res, const uint32_t* level_offset,
                                    const uint32_t* level_size, const float* d_w1, const float* d_b1, const float* d_w2,
                                    const float* d_b2, int hidden, float bound, float blob_density, float blob_radius,
                                    float sigmoid_saturation, float* d_sigmas, float* d_rgbs, void* stream);

/* Decoder backward (the reconstruct step): gradients of  sum(g_sigma * sigma) + sum(g_rgb * rgb)  for the M
 * points of a training batch w.r.t. the hash table and the MLP of iNGPDecoder.point_decode (lib/models/decoders/ingp_decoder.py:106-120;
 * sigma uses the backward of the reference's _trunc_exp, lib/ops/activation.py:17-20).  The forward is recomputed, nothing has to be
 * saved.  d_grad_table [rows][2] is ACCUMULATED into (float atomics; zero it per optimiser step); the four MLP gradients are
 * overwritten (deterministic two-stage reduction).  d_grad_rgb may be NULL (density-only).  hidden must be 64. */
MVE_API size_t mve_hashgrid_mlp_backward_workspace_bytes(uint32_t M, int n_levels);
MVE_API int mve_hashgrid_mlp_backward(const float* d_xyz, uint32_t M, const float* d_table, int n_levels, const float* level_scale,
                                      const uint32_t* level_res, const uint32_t* level_offset, const uint32_t* level_size,
                                      const float* d_w1, const float* d_b1, const float* d_w2, const float* d_b2, int hidden, float bound,
                                      float blob_density, float blob_radius, float sigmoid_saturation, const float* d_grad_sigma,
                                      const float* d_grad_rgb, float* d_grad_table, float* d_grad_w1, float* d_grad_b1,
                                      float* d_grad_w2, float* d_grad_b2, void* d_workspace, size_t workspace_bytes, void* stream);
/* torch.optim.Adam step (no weight decay, no amsgrad) in place on param / exp_avg / exp_avg_sq; step counts from 1. */
MVE_API int mve_adam_step(float* d_param, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, size_t n, float lr, float beta1,
                          float beta2, float eps, int step, void* stream);

/* VolumeRenderer.forward, eval branch (lib/models/decoders/base_volume_renderer.py:264-329) as ONE launch:
 * near/far from the aabb, occupancy-grid march (cascade count 1, no contraction, noise 0), hash-grid + MLP decode and
 * compositing per ray, with the reference's termination rules.  Outputs [N], [N], [N,3]; d_n_samples [N] optional. */
MVE_API int mve_nerf_render_rays(const float* d_rays_o, const float* d_rays_d, uint32_t N, const uint8_t* d_bitfield,
                                 uint32_t grid_size, const float* d_aabb, float bound, float min_near, float dt_gamma,
                                 uint32_t max_steps, float T_thresh, const float* d_table, int n_levels,
                                 const float* level_scale, const uint32_t* level_res, const uint32_t* level_offset,
                                 const uint32_t* level_size, const float* d_w1, const float* d_b1, const float* d_w2,
                                 const float* d_b2, int hidden, float blob_density, float blob_radius,
                                 float sigmoid_saturation, float* d_weights_sum, float* d_depth, float* d_image,
                                 int32_t* d_n_samples, void* stream);

/* get_ray_directions + get_rays(norm=True) (lib/core/utils/geometry_utils.py:18-55): intrinsics [V,4] (fx,fy,cx,cy),
 * poses [V,3,4] c2w -> rays_o, rays_d [V*h*w,3]; d_dir_norm [V*h*w] = |camera-space direction| (1/r -> 1/z factor), optional. */
MVE_API int mve_camera_rays(const float* d_intrinsics, const float* d_poses, int n_views, int h, int w, float* d_rays_o,
                            float* d_rays_d, float* d_dir_norm, void* stream);

/* depth_to_normal (geometry_utils.py:119-148, format 'opengl') fused with the tail of BaseNeRF.render
 * (lib/models/autoencoders/base_nerf.py:549-553): depth is 1/z; with d_alpha (element stride alpha_stride) the depth is
 * first divided by clamp(alpha,1e-6) and d_normal = normal_fg*alpha + normal_bg*(1-alpha).  normal_bg3: host [3] or NULL. */
MVE_API int mve_depth_to_normal(const float* d_depth, const float* d_alpha, int alpha_stride, const float* d_intrinsics,
                                int n_views, int h, int w, const float* normal_bg3, float* d_normal_fg, float* d_normal,
                                void* stream);

/* normalize_depth (geometry_utils.py:151-168): depths, alphas [V][hw] -> out [V][hw] */
MVE_API int mve_normalize_depth(const float* d_depths, const float* d_alphas, int n_views, int hw, float far_depth,
                                float alpha_clip, float eps, float* d_out, void* stream);

/* =========================================================================
 * 5. Texture-space ops of the mesh path.
 * ========================================================================= */

/* edge_dilation (lib/ops/edge_dilation.py:5-47): img [n,c,h,w] f32, mask [n,1,h,w] f32 -> dilated img (and mask) after
 * `iters` rounds of nearest-valid-texel fill inside a (2*round(radius)+1)^2 window.  *_tmp: ping-pong scratch of the same
 * sizes (may be NULL when iters <= 1).  Inputs are not modified. */
MVE_API int mve_edge_dilation(const float* d_img, const float* d_mask, int n, int c, int h, int w, float radius, int iters,
                              float* d_img_out, float* d_mask_out, float* d_img_tmp, float* d_mask_tmp, void* stream);

/* =========================================================================
 * 6. Mesh rasterisation (output convention of nvdiffrast as used at
 *    lib/models/decoders/mesh_renderer/base_mesh_renderer.py:240-252; coverage rules specified in oracle/raster_oracle.c).
 * ========================================================================= */

/* dr.rasterize(glctx, pos, tri, (H, W)) as called at base_mesh_renderer.py:240-241, :521, :543:
 * pos: [B][V][4] clip-space f32, tri: [F][3] i32 -> rast [B][H][W][4] = (u, v, z/w, triangle_id+1), 0 where empty.
 * Row 0 is y_ndc = -1 (OpenGL orientation).  d_workspace: >= mve_rasterize_workspace_bytes(B,H,W,F). */
MVE_API size_t mve_rasterize_workspace_bytes(int B, int H, int W, int F);
MVE_API int mve_rasterize(const float* d_pos, int B, int V, const int32_t* d_tri, int F, int H, int W, float* d_rast,
                          void* d_workspace, size_t workspace_bytes, void* stream);
/* dr.interpolate(attr, rast, tri)[0] (base_mesh_renderer.py:246-252, :259, :265, :273, :545, :556, :573):
 * attr [Battr][Vattr][A] (Battr 1 broadcasts), tri [F][3] indexes attr; out [B][H][W][A], 0 where empty */
MVE_API int mve_interpolate(const float* d_attr, int Battr, int Vattr, int A, const float* d_rast, int B, int H, int W,
                            const int32_t* d_tri, int F, float* d_out, void* stream);

/* Multi-view texture back-projection, the device part of MeshRenderer.bake_multiview
 * (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:507-603).  Bilinear filter with wrap addressing; the reference's
 * default 'linear-mipmap-linear' (nvdiffrast) is not reproduced -- see DESIGN.md.
 *   splat_visibility : sum of bilinear footprint weights of every foreground pixel (texc [n,h,w,2] uv, rast [n,h,w,4]) per
 *                      texel = d sum(texture(ones, texc)) / d ones (:547-552); d_vis_u64 [n][map][map] 2^-32 fixed point.
 *   view_weight      : clamp(-normal.dir, 0)^pow * alpha from the 1/z depth map, then the 5x5 min-pool (:554-566);
 *                      d_tmp and d_out are [n,h,w] f32.
 *   bake_accumulate  : per texel and view, fetch (rgb, view weight) at the texel's projected position (v_img [n,V,2] =
 *                      clip.xy/w*0.5+0.5 interpolated over the UV-space raster tex_rast [map,map,4] with faces d_f) and
 *                      add (rgb*weight, weight), weight = view_weight * visibility, into d_accum [map,map,4] (:568-582).
 *   bake_finalize    : albedo = sum / max(weight, 1e-8) -> [3][map][map] (ready for mve_edge_dilation). */
MVE_API int mve_splat_visibility(const float* d_texc, const float* d_rast, int n, int h, int w, int map_size, void* d_vis_u64,
                                 void* stream);
MVE_API int mve_view_weight(const float* d_depth, const float* d_alpha, const float* d_intrinsics, int n, int h, int w,
                            float cos_weight_pow, float* d_tmp, float* d_out, void* stream);
MVE_API int mve_bake_accumulate(const float* d_tex_rast, const int32_t* d_f, int F, const float* d_v_img, int V,
                                const float* d_images, const float* d_w_img, const void* d_vis_u64, int n, int h, int w,
                                int map_size, float* d_accum, void* stream);
MVE_API int mve_bake_finalize(const float* d_accum, int map_size, float* d_albedo_chw, void* stream);

/* Remaining pieces of MeshRenderer.forward (base_mesh_renderer.py:207-395):
 *   edge_opposites   : per triangle edge e = (tri[e], tri[(e+1)%3]) the vertex opposite to it in the one other triangle sharing the
 *                      edge, -1 for boundary / non-manifold edges (what dr.antialias derives from the topology) -> opp [F,3];
 *   antialias        : dr.antialias(color [B,h,w,C], rast, pos [B,V,4], tri) (:289-293) with the silhouette rules specified in
 *                      oracle/raster_oracle.c; d_out must not alias d_color;
 *   texture_bilinear : dr.texture(tex [Bt,th,tw,C] (Bt = 1 broadcasts), uv [n,h,w,2]) with the bilinear filter and wrap addressing;
 *                      pixels with rast[...,3] == 0 are written as 0 when d_rast is given (:258-263);
 *   box_downsample   : F.interpolate(mode='area', scale_factor=1/factor) on channel-last images (interpolate_hwc, :15-19, :380-383). */
MVE_API size_t mve_edge_opposites_workspace_bytes(int F);
MVE_API int mve_edge_opposites(const int32_t* d_tri, int F, int32_t* d_opp, void* d_workspace, size_t workspace_bytes, void* stream);
MVE_API int mve_antialias(const float* d_color, int B, int H, int W, int C, const float* d_rast, const float* d_pos, int V,
                          const int32_t* d_tri, int F, const int32_t* d_opp, float* d_out, void* stream);
MVE_API int mve_texture_bilinear(const float* d_tex, int Bt, int th, int tw, int C, const float* d_uv, const float* d_rast, int n, int h,
                                 int w, float* d_out, void* stream);
MVE_API int mve_box_downsample(const float* d_x, int B, int H, int W, int C, int factor, float* d_y, void* stream);

/* Marching tetrahedra, DMTet.__call__ (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:140-188): pos [Nv,3] f32,
 * sdf [Nv] f32 (occupied where sdf > 0), tets [Nt,4] int32 -> verts [n_verts,3] f32 (one per sign-changing edge, in the
 * lexicographic order of the reference's torch.unique), faces [n_faces,3] int32 (one-triangle tets first, then two-triangle
 * tets, each in tet order).  Two calls because the sizes are data dependent: count writes (n_verts, n_faces) to d_counts
 * (device int32[2]); after reading them the caller allocates the outputs and calls write with the SAME untouched workspace. */
MVE_API size_t mve_dmtet_workspace_bytes(size_t n_verts, size_t n_tets);
MVE_API int mve_dmtet_count(const float* d_sdf, const int32_t* d_tets, size_t n_verts, size_t n_tets, int32_t* d_counts,
                            void* d_workspace, size_t workspace_bytes, void* stream);
MVE_API int mve_dmtet_write(const float* d_pos, const float* d_sdf, const int32_t* d_tets, size_t n_verts, size_t n_tets,
                            float* d_out_verts, int32_t* d_out_faces, int32_t* d_out_edges /* optional [n_out,2]: the grid edge
                            (a < b) behind every output vertex, what the backward needs */, void* d_workspace,
                            size_t workspace_bytes, void* stream);
/* Backward of the vertex interpolation (the part of DMTet.__call__ that carries gradients, base_mesh_renderer.py:167-176):
 * d_grad_pos [Nv,3] and d_grad_sdf [Nv] are ACCUMULATED into (float atomics; either may be NULL). */
MVE_API int mve_dmtet_backward(const float* d_pos, const float* d_sdf, const int32_t* d_edges, size_t n_out_verts,
                               const float* d_grad_verts, float* d_grad_pos, float* d_grad_sdf, void* stream);

/* Backward of the render ops w.r.t. their colour-like input (mesh half): what nvdiffrast's autograd supplies
 * when the reference optimises textures / vertex colours through MeshRenderer.forward (lib/pipelines/mvedit_3d_pipeline.py:716-847,
 * lib/pipelines/mvedit_texture_pipeline.py optimisation loop).  The forward ops are linear in
 * that input, so these are their exact transposes; geometry (rast, pos) carries no gradient.
 *   interpolate_backward      : d_grad_attr [Battr,V,A] += barycentric weights * d_grad_out [B,h,w,A]   (accumulates; float atomics)
 *   texture_bilinear_backward : d_grad_tex [Bt,th,tw,C] += bilinear weights * d_grad_out [n,h,w,C]      (accumulates; float atomics)
 *   antialias_backward        : d_grad_color [B,h,w,C] = transpose of mve_antialias applied to d_grad_out (overwrites; deterministic) */
MVE_API int mve_interpolate_backward(const float* d_grad_out, int Battr, int Vattr, int A, const float* d_rast, int B, int H, int W,
                                     const int32_t* d_tri, int F, float* d_grad_attr, void* stream);
MVE_API int mve_texture_bilinear_backward(const float* d_grad_out, int Bt, int th, int tw, int C, const float* d_uv, const float* d_rast,
                                          int n, int h, int w, float* d_grad_tex, void* stream);
MVE_API int mve_antialias_backward(const float* d_grad_out, int B, int H, int W, int C, const float* d_rast, const float* d_pos, int V,
                                   const int32_t* d_tri, int F, const int32_t* d_opp, float* d_grad_color, void* stream);

/* Tri-plane radiance decoders: `TriPlaneDecoder.point_decode` (lib/models/decoders/triplane_decoder.py:135-199)
 * and `TriPlaneiNGPDecoder.point_decode` (lib/models/decoders/triplane_ingp_decoder.py:142-212) for one scene, forward (backward below):
 *   feature k = c * 3 + p of a point = bilinear F.grid_sample(padding_mode='border', align_corners=False) of channel c of plane p at the
 *   point's two coordinates for that plane (plane_cfg; flip_z negates z);  base_x = base_net(features) [+ ingp_base_net(hash-grid encoding
 *   of (xyz + bound) / (2 bound), tiny-cuda-nn HashGrid with Smoothstep interpolation)];  sigma = sigma_activation(density_net(act(base_x)));
 *   rgb = sigmoid(color_net(cat[act(base_x), SH_4(dir)])) * (1 + 2 s) - s.
 * Supported topology (the classes' defaults): base_net = one Linear(3C -> hidden), density_net = Linear(hidden -> 1), color_net =
 * Linear(hidden + 16 -> hidden2), act, Linear(hidden2 -> 3); hidden, hidden2 in {64, 128}; dir_layers = None; ingp_base_layers = 1.
 * Weight matrices are passed TRANSPOSED ([in][out], fp32) so that a wave reads an input's fan-out as one contiguous scalar load.
 * code is channels-last [3][h][w][C].  activation: 0 relu, 1 silu, 2 softplus; sigma_activation: the same, 3 = trunc_exp (exp). */
typedef struct MveTriplaneDesc {
    const float *d_xyz, *d_dirs;                     /* [N,3]; d_dirs NULL: density only (point_density_decode) */
    const float *d_code;                             /* [3][h][w][C] */
    int32_t N, C, h, w;
    int32_t axes[6];                                 /* plane p reads (xyz[axes[2p]], xyz[axes[2p+1]]) */
    int32_t flip_z;
    const float *d_base_wT, *d_base_b;               /* [3C][hidden], [hidden] */
    int32_t hidden, hidden2;
    const float *d_ingp_wT, *d_ingp_b;               /* [2 n_levels][hidden], [hidden]; NULL for TriPlaneDecoder */
    const float *d_table;                            /* hash table [rows][2] */
    int32_t n_levels;
    float bound;
    const float *level_scale;                        /* HOST arrays [n_levels] (as mve_hashgrid_mlp_decode takes them) */
    const uint32_t *level_res, *level_offset, *level_size;
    const float *d_dens_w, *d_dens_b;                /* [hidden], [1] */
    const float *d_col1_wT, *d_col1_b;               /* [hidden + 16][hidden2], [hidden2] */
    const float *d_col2_w, *d_col2_b;                /* [3][hidden2], [3] */
    int32_t activation, sigma_activation;
    float sigmoid_saturation;
    float *d_sigmas, *d_rgbs;                        /* [N], [N,3] (d_rgbs unused when d_dirs is NULL) */
} MveTriplaneDesc;
MVE_API int mve_triplane_decode(const MveTriplaneDesc* desc, void* stream);
/* Backward of mve_triplane_decode: gradients of sum(d_grad_sigmas * sigma) + sum(d_grad_rgbs * rgb) (d_grad_rgbs NULL: density only) -- what
 * autograd supplies when nerf_optim optimises a TriPlaneiNGPDecoder scene (lib/pipelines/mvedit_3d_pipeline.py:507-633 through
 * triplane_ingp_decoder.py:142-212; trunc_exp backward as lib/ops/activation.py:17-20).  Weight gradients come in the torch layout of the
 * reference modules ([out][in], OVERWRITTEN, reduced in a fixed order); d_code [3][h][w][C] and d_table [rows][2] are ACCUMULATED into
 * (float atomics, like F.grid_sample's / tiny-cuda-nn's backward; either may be NULL).  The forward is recomputed: desc's outputs are not
 * read or written. */
typedef struct MveTriplaneGrads {
    float *d_code, *d_table;
    float *d_base_w, *d_base_b;                      /* [hidden][3C], [hidden] */
    float *d_ingp_w, *d_ingp_b;                      /* [hidden][2 n_levels], [hidden] (NULL without a hash branch) */
    float *d_dens_w, *d_dens_b;                      /* [1][hidden], [1] */
    float *d_col1_w, *d_col1_b;                      /* [hidden2][hidden + 16], [hidden2] */
    float *d_col2_w, *d_col2_b;                      /* [3][hidden2], [3] */
} MveTriplaneGrads;
MVE_API size_t mve_triplane_backward_workspace_bytes(int N, int C, int hidden, int hidden2, int n_levels);
MVE_API int mve_triplane_backward(const MveTriplaneDesc* desc, const float* d_grad_sigmas, const float* d_grad_rgbs, const MveTriplaneGrads* grads,
                                  void* d_workspace, size_t workspace_bytes, void* stream);

/* Mip-mapped texture path: what the reference gets from nvdiffrast with MeshRenderer(texture_filter='linear-mipmap-linear') -- its default
 * (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:196) -- in forward (:241, :260-264, :357-361), get_cam_weights_uv (:442, :466-475,
 * :496-500) and bake_multiview (:521, :543-552, :573-577).  nvdiffrast is not vendored: algorithm restated in oracle/texture_mip_oracle.py.
 *   rasterize_db        dr.rasterize(...)[1]: d_rast_db [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY) of the barycentrics, 0 on empty pixels
 *   interpolate_da      dr.interpolate(attr, rast, tri, rast_db=..., diff_attrs='all')[1]: d_out [B,npix,2C] = (dA0/dX, dA0/dY, dA1/dX, ...)
 *   mip_levels          number of levels above level 0 of the full stack (down to 1 x 1); mip_texels = texels of levels 1..max_level
 *   mip_build           d_mips [Bt][mip_texels*C]: level l+1 = 2x2 box filter of level l (extents must be even at every level, or 1)
 *   texture_mip         dr.texture(tex [Bt,H,W,C], uv [n,h,w,2], uv_da [n,h,w,4], filter_mode='linear-mipmap-linear'), wrap addressing;
 *                       Bt = 1 broadcasts; pixels with rast[..,3] == 0 give 0 when d_rast is given (albedo[~fg] = 0, :264); C <= 8
 *   texture_mip_backward  gradient w.r.t. the texture: overwrites d_g_tex0 [Bt,H,W,C]; d_g_mips is scratch of the mip stack's size
 *   visibility_mip      `visibility_grad` (:470-475, :547-552): d sum(dr.texture(ones [n,map,map,1], texc, uv_da)) / d ones per view, accumulated
 *                       in 2^-32 fixed point (order independent) -> d_vis [n,map,map] fp32
 *   bake_accumulate_mip per texel and view: mip-mapped fetch of (rgb, view weight) from d_img0 [n,h,w,4] (+ its mip stack) at the texel's
 *                       projection, times d_vis; d_accum [map,map,4] += (rgb * weight, weight)   (:573-582) */
MVE_API int mve_rasterize_db(const float* d_pos, int B, int V, const int32_t* d_tri, int F, const float* d_rast, int H, int W,
                             float* d_rast_db, void* stream);
MVE_API int mve_interpolate_da(const float* d_attr, int attr_batch, int V, int C, const float* d_rast, const float* d_rast_db, int B, int npix,
                               const int32_t* d_tri, int F, float* d_out, void* stream);
MVE_API int mve_mip_levels(int H, int W);
MVE_API size_t mve_mip_texels(int H, int W, int max_level);
MVE_API int mve_mip_build(const float* d_tex0, int Bt, int H, int W, int C, int max_level, float* d_mips, void* stream);
MVE_API int mve_texture_mip(const float* d_tex0, const float* d_mips, int Bt, int H, int W, int C, int max_level, const float* d_uv,
                            const float* d_uv_da, const float* d_rast, int n, int h, int w, float* d_out, void* stream);
MVE_API int mve_texture_mip_backward(const float* d_g_out, int Bt, int H, int W, int C, int max_level, const float* d_uv,
                                     const float* d_uv_da, const float* d_rast, int n, int h, int w, float* d_g_tex0, float* d_g_mips,
                                     void* stream);
MVE_API size_t mve_visibility_mip_workspace_bytes(int n, int map_size, int max_level);
MVE_API int mve_visibility_mip(const float* d_texc, const float* d_texc_da, const float* d_rast, int n, int h, int w, int map_size,
                               int max_level, void* d_workspace, size_t workspace_bytes, float* d_vis, void* stream);
MVE_API int mve_bake_accumulate_mip(const float* d_tex_rast, const float* d_tex_rast_db, const int32_t* d_f, int F, const float* d_v_img,
                                    int V, const float* d_img0, const float* d_img_mips, int h, int w, int max_level, const float* d_vis,
                                    int n, int map_size, float* d_accum, void* stream);

/* Geometry gradients of dr.rasterize / dr.interpolate (nvdiffrast backward as the reference's mesh optimisation uses it through
 * MeshRenderer.forward, base_mesh_renderer.py:240-252):
 *   interpolate_backward_rast: grad_rast[pixel] = (sum_a g_a (a0 - a2), sum_a g_a (a1 - a2), 0, 0)   -- d out / d (u, v)
 *   rasterize_backward       : grad_pos[b][vertex] += d (u, v, z/w)[pixel] / d clip-space vertex, every pixel's triangle held fixed
 *                              (continuous barycentrics; float atomics; the caller zero-fills d_grad_pos [B][V][4]). */
MVE_API int mve_interpolate_backward_rast(const float* d_attr, int Battr, int Vattr, int A, const float* d_rast, int B, int H, int W,
                                          const int32_t* d_tri, int F, const float* d_grad_out, float* d_grad_rast, void* stream);
MVE_API int mve_rasterize_backward(const float* d_pos, int B, int V, const int32_t* d_tri, int F, int H, int W, const float* d_rast,
                                   const float* d_grad_rast, float* d_grad_pos, void* stream);
/* dr.antialias backward w.r.t. the clip-space vertices (the silhouette gradient): with the pair decisions of the forward held fixed,
 * out[dst] = c[dst] + |tt - 1/2| (c[src] - c[dst]) and tt is a smooth function of the crossed edge's two vertices.  grad_pos [B][V][4]
 * is accumulated with float atomics (caller zero-fills or chains it after mve_rasterize_backward). */
MVE_API int mve_antialias_backward_pos(const float* d_color, const float* d_grad_out, int B, int H, int W, int C, const float* d_rast,
                                       const float* d_pos, int V, const int32_t* d_tri, int F, const int32_t* d_opp, float* d_grad_pos,
                                       void* stream);

/* TRACER-B7 foreground segmentor (lib/models/segmentors/tracer_b7.py:16-73, called once per denoise step from
 * lib/pipelines/adapter3d_mixin.py:14-19; EfficientNet-B7 encoder lib/models/architecture/tracerb7/efficientnet.py + TRACER decoder
 * tracer.py / att_modules.py / conv_modules.py): the operators the GEMM / 3x3-conv entry points do not cover.  Activations are NHWC
 * 16-bit ([B, H, W, C] contiguous: a 1x1 convolution is mve_gemm on [B*H*W, C]); single-channel decoder maps are fp32; BatchNorm
 * arrives folded into weights + bias (mvedit_amd/segmentor.py folds it when the state dict is loaded).  act: 0 none, 1 swish,
 * 2 SELU, 3 ReLU, 4 sigmoid.
 *   seg_conv2d      : nn.Conv2d with groups = C (depthwise = 1: weights [kh][kw][C] f32; the _depthwise_conv of MBConvBlock,
 *                     efficientnet.py:66-75, with the STATIC TensorFlow-"SAME" padding of effi_utils.py:270-315 passed as explicit
 *                     top / left padding, and DWConv / DWSConv of conv_modules.py:42-88) or groups = 1 (weights [Cout][kh][kw][Cin] f32:
 *                     BasicConv2d with 1xk / kx1 / dilated kernels, conv_modules.py:11-39, att_modules.py:16-41).  out = act(conv + bias)
 *                     [* mul] [+ add]; x / out / mul / add rows are ldx / ldo / ld2 channels wide (channel slices of wider tensors).
 *   seg_act         : in-place activation of a GEMM / conv3x3 output (the swish / SELU behind a BatchNorm).
 *   seg_mconv       : small dense convolutions (stride 1, output size = input size) on the matrix cores with everything around them fused:
 *                     out[m][n] = act(sum_{tap, c} (x[pixel(m) + tap][c] * gate[b][c]) W[n][tap][c] + bias[n]) (+ residual[m][n]); W [N][ldw]
 *                     dtype with rows [kh][kw][Cin]; gate [B][Cin] f32 or NULL (1 x 1 only: the squeeze-and-excite scaling of
 *                     MBConvBlock.forward, efficientnet.py:124-131, applied to the operand in registers).  The expand conv + swish (:110-113),
 *                     the project conv + skip (:131-141), and the RFB block's 1 x 1, 1 x k, k x 1 and dilated 3 x 3 BasicConv2d layers
 *                     (conv_modules.py / att_modules.py:23-72; Cin % 32 == 0 when the kernel has taps).  Alignment: d_x, d_w 16 bytes;
 *                     d_out / d_residual 8 bytes, and 16 when ldo / ldr are multiples of 8 (rows are then stored 32 bytes per lane).
 *   seg_channel_mean: F.adaptive_avg_pool2d(x, 1) / GlobalAvgPool -> f32 [B][C].
 *   seg_se_gate     : sigmoid(_se_expand(swish(_se_reduce(pooled)))) (efficientnet.py:124-129), w1 [S][C], w2 [C][S] f32; pooled[b][c] = scale *
 *                     sum_k sums[b][k][c] (nslab = 1, scale = 1: a pooled vector as it is); d_hidden: [B][S + C] f32 workspace.
 *   seg_dwconv_pool : the MBConv block's depthwise convolution + swish (efficientnet.py:115-121) that also leaves the per-channel sums of its
 *                     (rounded) output per pixel slab, d_sums [B][seg_dwconv_slabs(B, Ho, Wo, C, k, stride)][C] f32 -- the squeeze without a second pass.
 *   seg_scale       : x = src * A[b][c] (+ S[b][c]) -- the SE gating; UnionAttentionModule's x * att + x, BatchNorm and confidence mask.
 *   seg_resize      : bilinear F.interpolate / torchvision Resize(antialias=False) (align_corners flag); in_mode 0: NHWC dtype, 1: NHWC
 *                     f32, 2: NCHW f32; optional (x - mean[c]) / std[c] (transforms.Normalize, tracer_b7.py:39-44).
 *   seg_uam_channel : UnionAttentionModule.channel_tracer + masking (att_modules.py:135-168) on the pooled vector; C <= 256.  Outputs the
 *                     sigmoid gate att and the per-(image, channel) affine A, S with BatchNorm(x * att + x) * mask = x * A + S (:163-174).
 *   seg_mul         : elementwise x * y (* z) (Aggregation.forward's gating products, att_modules.py:226-233).
 *   seg_uam_spatial : the spatial SDPA of UnionAttentionModule.forward (:176-187) over qkv [B][H*W][3] f32.
 *   seg_object_mix  : ObjectAttention's masked encoder map (att_modules.py:277-282).
 *   seg_fuse        : sigmoid((up4(d2) + up8(d1) + up8(d0)) / 3) (tracer.py:86-97).
 *   seg_post        : -max_pool2d(-m, 2 e + 1), resize to [Ho][Wo], failure rule (tracer_b7.py:66-72); out [B][Ho][Wo] dtype or f32. */
MVE_API int mve_seg_conv2d(int dtype, const void* d_x, int B, int H, int W, int Cin, int ldx, const float* d_w, const float* d_bias, void* d_out,
                           int Ho, int Wo, int Cout, int ldo, int kh, int kw, int stride, int pad_t, int pad_l, int dil, int depthwise, int act,
                           const void* d_mul, const void* d_add, int ld2, int out_f32, void* stream);
MVE_API int mve_seg_mconv(int dtype, const void* d_x, int B, int H, int W, int Cin, int ldx, const void* d_w, int ldw, int kh, int kw, int dil,
                          int pad_t, int pad_l, const float* d_bias, const float* d_gate, const void* d_residual, int ldr, void* d_out, int N,
                          int ldo, int act, void* stream);
MVE_API int mve_seg_act(int dtype, void* d_x, size_t n, int act, void* stream);
MVE_API int mve_seg_channel_mean(int dtype, const void* d_x, int B, int HW, int C, float* d_out, void* stream);
MVE_API int mve_seg_se_gate(const float* d_sums, int nslab, float scale, int B, int C, int S, const float* d_w1, const float* d_b1, const float* d_w2,
                            const float* d_b2, float* d_hidden, float* d_gate, void* stream);
MVE_API int mve_seg_dwconv_slabs(int B, int Ho, int Wo, int C, int k, int stride);
MVE_API int mve_seg_dwconv_pool(int dtype, const void* d_x, int B, int H, int W, int C, int ldx, const float* d_w, const float* d_bias, void* d_out, int Ho,
                                int Wo, int ldo, int kh, int kw, int stride, int pad_t, int pad_l, int act, float* d_sums, void* stream);
MVE_API int mve_seg_scale(int dtype, void* d_x, const void* d_src, int B, int HW, int C, const float* d_A, const float* d_S, void* stream);
MVE_API int mve_seg_resize(int dtype, const void* d_x, int B, int H, int W, int C, void* d_out, int Ho, int Wo, int align_corners, int in_mode,
                           int out_f32, const float* d_mean, const float* d_std, void* stream);
MVE_API int mve_seg_uam_channel(const float* d_pooled, int B, int C, const float* d_ns, const float* d_nb, const float* d_wq, const float* d_wk,
                                const float* d_wv, const float* d_wfc, float ratio, const float* d_bn_scale, const float* d_bn_shift, float* d_att,
                                float* d_A, float* d_S, void* stream);
MVE_API int mve_seg_mul(int dtype, const void* d_x, const void* d_y, const void* d_z, void* d_out, size_t n, void* stream);
MVE_API int mve_seg_uam_spatial(const float* d_qkv, int B, int H, int W, float* d_out, void* stream);
MVE_API int mve_seg_object_mix(int dtype, const float* d_map, const void* d_enc, void* d_out, int B, int HW, int C, void* stream);
MVE_API int mve_seg_fuse(const float* d_d0, const float* d_d1, const float* d_d2, int B, int Hs, int Ws, float* d_out, void* stream);
MVE_API size_t mve_seg_post_workspace_bytes(int B, int Hs, int Ws, int Ho, int Wo);
MVE_API int mve_seg_post(int dtype, const float* d_m, int B, int Hs, int Ws, int erosion, void* d_out, int Ho, int Wo, int out_f32,
                         void* d_workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVEDIT_AMD_H */
