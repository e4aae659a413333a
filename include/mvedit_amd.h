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
MVE_API int mve_version(void);                 /* ABI version, bumped on breaking change */
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

#ifdef __cplusplus
}
#endif
#endif /* MVEDIT_AMD_H */
