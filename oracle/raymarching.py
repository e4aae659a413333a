"""ORACLE -- test infrastructure only: numpy front-end of raymarching_oracle.c.

Signatures mirror lib/ops/raymarching/raymarching.py of the reference (numpy arrays instead of
CUDA tensors).  Used as the checker for mvedit_amd.raymarching; never imported by the product.
"""
import ctypes

import numpy as np

from . import build

_lib = ctypes.CDLL(build())
_f = ctypes.c_float
_u = ctypes.c_uint32
_i = ctypes.c_int
_p = ctypes.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_lib.orc_march_rays_train_count.restype = _u
for name, args in {
    'orc_near_far_from_aabb': [_p, _p, _p, _u, _f, _p, _p],
    'orc_morton3d': [_p, _u, _p],
    'orc_morton3d_invert': [_p, _u, _p],
    'orc_packbits': [_p, _u, _f, _p],
    'orc_flatten_rays': [_p, _u, _u, _p],
    'orc_march_rays_train_count': [_p, _p, _p, _f, _i, _f, _u, _u, _u, _u, _p, _p, _p, _p],
    'orc_march_rays_train_write': [_p, _p, _p, _f, _i, _f, _u, _u, _u, _u, _p, _p, _p, _p, _p, _p, _p],
    'orc_composite_rays_train_forward': [_p, _p, _p, _p, _u, _u, _f, _i, _p, _p, _p, _p],
    'orc_composite_rays_train_backward': [_p] * 11 + [_u, _u, _f, _i, _p, _p],
    'orc_march_rays': [_u, _u, _p, _p, _p, _p, _f, _i, _f, _u, _u, _u, _p, _p, _p, _p, _p, _p, _p],
    'orc_composite_rays': [_u, _u, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p],
}.items():
    getattr(_lib, name).argtypes = args


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    _lib.orc_near_far_from_aabb(_ptr(rays_o), _ptr(rays_d), _ptr(aabb), N, min_near, _ptr(nears), _ptr(fars))
    return nears, fars


def morton3D(coords):
    coords = _i32(coords)
    out = np.empty(coords.shape[0], np.int32)
    _lib.orc_morton3d(_ptr(coords), coords.shape[0], _ptr(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices)
    out = np.empty((indices.shape[0], 3), np.int32)
    _lib.orc_morton3d_invert(_ptr(indices), indices.shape[0], _ptr(out))
    return out


def packbits(grid, thresh):
    grid = _f32(grid)
    N = grid.size // 8
    out = np.empty(N, np.uint8)
    _lib.orc_packbits(_ptr(grid), N, thresh, _ptr(out))
    return out


def flatten_rays(rays, M):
    rays = _i32(rays)
    res = np.zeros(M, np.int32)
    _lib.orc_flatten_rays(_ptr(rays), rays.shape[0], M, _ptr(res))
    return res


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, noises, dt_gamma=0, max_steps=1024,
                     contract=False):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    grid = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    nears, fars, noises = _f32(nears), _f32(fars), _f32(noises)
    N = rays_o.shape[0]
    rays = np.empty((N, 2), np.int32)
    common = (_ptr(rays_o), _ptr(rays_d), _ptr(grid), bound, int(contract), dt_gamma, max_steps, N, C, H, _ptr(nears),
              _ptr(fars), _ptr(noises))
    M = _lib.orc_march_rays_train_count(*common, _ptr(rays))
    xyzs, dirs, ts = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    _lib.orc_march_rays_train_write(*common, _ptr(rays), _ptr(xyzs), _ptr(dirs), _ptr(ts))
    return xyzs, dirs, ts, rays


def composite_rays_train(sigmas, rgbs, ts, rays, T_thresh=1e-4, binarize=False):
    sigmas, rgbs, ts, rays = _f32(sigmas), _f32(rgbs), _f32(ts), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    weights = np.zeros(M, np.float32)
    weights_sum, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    _lib.orc_composite_rays_train_forward(_ptr(sigmas), _ptr(rgbs), _ptr(ts), _ptr(rays), M, N, T_thresh, int(binarize),
                                          _ptr(weights), _ptr(weights_sum), _ptr(depth), _ptr(image))
    return weights, weights_sum, depth, image


def composite_rays_train_backward(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                  weights_sum, depth, image, T_thresh=1e-4, binarize=False):
    a = [_f32(x) for x in (grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts)]
    rays = _i32(rays)
    b = [_f32(x) for x in (weights_sum, depth, image)]
    M, N = a[4].shape[0], rays.shape[0]
    grad_sigmas, grad_rgbs = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    _lib.orc_composite_rays_train_backward(*[_ptr(x) for x in a], _ptr(rays), *[_ptr(x) for x in b], M, N, T_thresh,
                                           int(binarize), _ptr(grad_sigmas), _ptr(grad_rgbs))
    return grad_sigmas, grad_rgbs


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, noises,
               dt_gamma=0, max_steps=1024, contract=False):
    rays_alive, rays_t = _i32(rays_alive), _f32(rays_t)
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    grid = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    near, far, noises = _f32(near), _f32(far), _f32(noises)
    M = n_alive * n_step
    xyzs, dirs, ts = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    _lib.orc_march_rays(n_alive, n_step, _ptr(rays_alive), _ptr(rays_t), _ptr(rays_o), _ptr(rays_d), bound, int(contract),
                        dt_gamma, max_steps, C, H, _ptr(grid), _ptr(near), _ptr(far), _ptr(xyzs), _ptr(dirs), _ptr(ts),
                        _ptr(noises))
    return xyzs, dirs, ts


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh=1e-2,
                   binarize=False):
    """In place on the numpy arrays rays_alive (int32), rays_t, weights_sum, depth, image (float32, contiguous)."""
    for a in (rays_t, weights_sum, depth, image):
        assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    assert rays_alive.dtype == np.int32 and rays_alive.flags['C_CONTIGUOUS']
    sigmas, rgbs, ts = _f32(sigmas), _f32(rgbs), _f32(ts)
    _lib.orc_composite_rays(n_alive, n_step, T_thresh, int(binarize), _ptr(rays_alive), _ptr(rays_t), _ptr(sigmas),
                            _ptr(rgbs), _ptr(ts), _ptr(weights_sum), _ptr(depth), _ptr(image))
