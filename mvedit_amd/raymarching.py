"""Drop-in mirror of the reference's `lib.ops.raymarching` operator interface
(lib/ops/raymarching/__init__.py:1-8, raymarching.py) on top of libmvedit_amd.

Same function names, argument order/meaning and return values; tensors are
PyTorch-ROCm tensors used as storage only, arithmetic runs in the HIP kernels
of mvedit_amd/csrc/raymarching.hip through the C ABI.  As in the reference every
floating input is cast to fp32 (`custom_fwd(cast_inputs=torch.float32)`,
raymarching.py:33,99,180,239,315,437,495).

Differences a caller can observe (documented in DESIGN.md):
  * `march_rays_train` returns ray offsets that are an exclusive prefix sum in
    ray order (the reference's offsets depend on atomicAdd arrival order);
  * CPU tensors are moved to the current GPU like the reference does with
    `.cuda()`; there is no CPU execution path.
"""
from itertools import groupby

import torch
from torch.autograd import Function

from . import _lib

__all__ = ['near_far_from_aabb', 'sph_from_ray', 'morton3D', 'morton3D_invert', 'packbits', 'flatten_rays',
           'march_rays_train', 'composite_rays_train', 'march_rays', 'composite_rays',
           'batch_near_far_from_aabb', 'batch_composite_rays_train', 'compact_alive']


def _gpu_f32(x):
    if not x.is_cuda:
        x = x.cuda()
    return x.float().contiguous()


def _gpu(x):
    return x if x.is_cuda else x.cuda()


def _s(t):
    return _lib.stream_ptr(t.device)


# ----------------------------------------------------------------------------
# utils
# ----------------------------------------------------------------------------
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """rays_o/d [N,3], aabb [6] -> nears [N], fars [N]  (raymarching.py:31-65)."""
    rays_o = _gpu_f32(rays_o).view(-1, 3)
    rays_d = _gpu_f32(rays_d).view(-1, 3)
    aabb = _gpu_f32(aabb)
    N = rays_o.shape[0]
    nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    with torch.cuda.device(rays_o.device):
        _lib.call('mve_near_far_from_aabb', _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(aabb), N, float(min_near),
                  _lib.ptr(nears), _lib.ptr(fars), _s(rays_o))
    return nears, fars


def batch_near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching.py:71-97."""
    if isinstance(rays_o, torch.Tensor):
        assert rays_o.size() == rays_d.size()
        num_scenes, num_rays, _ = rays_o.size()
        nears, fars = near_far_from_aabb(rays_o.reshape(num_scenes * num_rays, 3),
                                         rays_d.reshape(num_scenes * num_rays, 3), aabb, min_near)
        return nears.reshape(num_scenes, num_rays), fars.reshape(num_scenes, num_rays)
    if len(rays_o) == 1:
        nears, fars = near_far_from_aabb(rays_o[0], rays_d[0], aabb, min_near)
        return [nears], [fars]
    counts = [r.size(0) for r in rays_o]
    nears, fars = near_far_from_aabb(torch.cat(rays_o, dim=0), torch.cat(rays_d, dim=0), aabb, min_near)
    return nears.split(counts), fars.split(counts)


def sph_from_ray(rays_o, rays_d, radius):
    """Background-sphere coordinates (raymarching.py:100-131).  The reference never reaches this
    on the MVEdit path (bg_radius=-1, base_volume_renderer.py:20); it is pure elementwise math and
    stays in torch."""
    rays_o = _gpu_f32(rays_o).view(-1, 3)
    rays_d = _gpu_f32(rays_d).view(-1, 3)
    A = (rays_d * rays_d).sum(-1)
    B = (rays_o * rays_d).sum(-1)
    Cc = (rays_o * rays_o).sum(-1) - radius * radius
    t = (-B + torch.sqrt(B * B - A * Cc)) / A
    p = rays_o + t[:, None] * rays_d
    theta = torch.atan2(torch.sqrt(p[:, 0] ** 2 + p[:, 2] ** 2), p[:, 1])
    phi = torch.atan2(p[:, 2], p[:, 0])
    return torch.stack([2 * theta / torch.pi - 1, phi / torch.pi], dim=-1)


def morton3D(coords):
    """coords [N,3] int32 -> indices [N] int32  (raymarching.py:134-155)."""
    coords = _gpu(coords).int().contiguous()
    N = coords.shape[0]
    indices = torch.empty(N, dtype=torch.int32, device=coords.device)
    with torch.cuda.device(coords.device):
        _lib.call('mve_morton3d', _lib.ptr(coords), N, _lib.ptr(indices), _s(coords))
    return indices


def morton3D_invert(indices):
    """indices [N] int32 -> coords [N,3] int32  (raymarching.py:161-181)."""
    indices = _gpu(indices).int().contiguous()
    N = indices.shape[0]
    coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    with torch.cuda.device(indices.device):
        _lib.call('mve_morton3d_invert', _lib.ptr(indices), N, _lib.ptr(coords), _s(indices))
    return coords


def packbits(grid, thresh, bitfield=None):
    """grid [C, H^3] f32 -> bitfield uint8 [C*H^3/8]  (raymarching.py:187-211)."""
    grid = _gpu_f32(grid)
    C, H3 = grid.shape[0], grid.shape[1]
    N = C * H3 // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    else:
        assert bitfield.is_cuda and bitfield.dtype == torch.uint8 and bitfield.is_contiguous() and bitfield.numel() >= N
    with torch.cuda.device(grid.device):
        _lib.call('mve_packbits', _lib.ptr(grid), N, float(thresh), _lib.ptr(bitfield), _s(grid))
    return bitfield


def flatten_rays(rays, M):
    """rays [N,2] (offset,count) -> res [M] ray index per sample  (raymarching.py:217-237)."""
    rays = _gpu(rays).int().contiguous()
    N = rays.shape[0]
    res = torch.zeros(M, dtype=torch.int32, device=rays.device)
    with torch.cuda.device(rays.device):
        _lib.call('mve_flatten_rays', _lib.ptr(rays), N, int(M), _lib.ptr(res), _s(rays))
    return res


# ----------------------------------------------------------------------------
# train
# ----------------------------------------------------------------------------
def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars,
                     perturb=False, dt_gamma=0, max_steps=1024, contract=False, noises=None):
    """-> xyzs [M,3], dirs [M,3], ts [M,2], rays [N,2] int32  (raymarching.py:243-311).

    `noises` (optional [N] tensor) overrides the internally drawn perturbation; tests use it to
    feed the same noise to the oracle.
    """
    rays_o = _gpu_f32(rays_o).view(-1, 3)
    rays_d = _gpu_f32(rays_d).view(-1, 3)
    dev = rays_o.device
    density_bitfield = _gpu(density_bitfield).contiguous()
    nears = _gpu_f32(nears)
    fars = _gpu_f32(fars)
    N = rays_o.shape[0]
    if noises is not None:
        noises = _gpu_f32(noises)
    elif perturb:
        noises = torch.rand(N, dtype=torch.float32, device=dev)
    else:
        noises = torch.zeros(N, dtype=torch.float32, device=dev)
    rays = torch.empty(N, 2, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(_lib.raw('mve_march_scratch_bytes')(N), dtype=torch.uint8, device=dev)
    common = (_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(density_bitfield), float(bound), int(bool(contract)),
              float(dt_gamma), int(max_steps), N, int(C), int(H), _lib.ptr(nears), _lib.ptr(fars), _lib.ptr(noises))
    with torch.cuda.device(dev):
        _lib.call('mve_march_rays_train_count', *common, _lib.ptr(rays), _lib.ptr(total), _lib.ptr(scratch), _s(rays_o))
        M = int(total.item())  # same single host read as the reference (raymarching.py:290)
        xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        ts = torch.zeros(M, 2, dtype=torch.float32, device=dev)
        _lib.call('mve_march_rays_train_write', *common, _lib.ptr(rays), M, _lib.ptr(xyzs), _lib.ptr(dirs),
                  _lib.ptr(ts), _s(rays_o))
    return xyzs, dirs, ts, rays


class _composite_rays_train(Function):
    """raymarching.py:314-372 (autograd Function with the analytic backward)."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, ts, rays, T_thresh=1e-4, binarize=False):
        sigmas = _gpu_f32(sigmas)
        rgbs = _gpu_f32(rgbs)
        ts = _gpu_f32(ts)
        rays = _gpu(rays).int().contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights = torch.zeros(M, dtype=torch.float32, device=dev)
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call('mve_composite_rays_train_forward', _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(ts), _lib.ptr(rays),
                      M, N, float(T_thresh), int(bool(binarize)), _lib.ptr(weights), _lib.ptr(weights_sum),
                      _lib.ptr(depth), _lib.ptr(image), _s(sigmas))
        ctx.save_for_backward(sigmas, rgbs, ts, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh, binarize]
        return weights, weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights, grad_weights_sum, grad_depth, grad_image):
        sigmas, rgbs, ts, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh, binarize = ctx.dims
        grad_weights = grad_weights.float().contiguous()
        grad_weights_sum = grad_weights_sum.float().contiguous()
        grad_depth = grad_depth.float().contiguous()
        grad_image = grad_image.float().contiguous()
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        with torch.cuda.device(sigmas.device):
            _lib.call('mve_composite_rays_train_backward', _lib.ptr(grad_weights), _lib.ptr(grad_weights_sum),
                      _lib.ptr(grad_depth), _lib.ptr(grad_image), _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(ts),
                      _lib.ptr(rays), _lib.ptr(weights_sum), _lib.ptr(depth), _lib.ptr(image), M, N, float(T_thresh),
                      int(bool(binarize)), _lib.ptr(grad_sigmas), _lib.ptr(grad_rgbs), _s(sigmas))
        return grad_sigmas, grad_rgbs, None, None, None, None


composite_rays_train = _composite_rays_train.apply


def _all_equal(iterable):
    g = groupby(iterable)
    return next(g, True) and not next(g, False)


def batch_composite_rays_train(sigmas, rgbs, ts, rays, num_points, T_thresh=1e-4, binarize=False):
    """raymarching.py:380-431: concatenate per-scene ray tables, shifting offsets."""
    num_scenes = len(ts)
    if num_scenes == 1:
        weights, weights_sum, depth, image = composite_rays_train(sigmas, rgbs, ts[0], rays[0], T_thresh, binarize)
        return weights, weights_sum[None], depth[None], image[None]
    ts_ = torch.cat(ts, dim=0)
    rays_, num_rays, shift = [], [], 0
    for ray_single, n_pts in zip(rays, num_points):
        rays_.append(torch.stack([ray_single[:, 0] + shift, ray_single[:, 1]], dim=-1))
        shift += n_pts
        num_rays.append(ray_single.size(0))
    rays_ = torch.cat(rays_, dim=0)
    weights, weights_sum_, depth_, image_ = composite_rays_train(sigmas, rgbs, ts_, rays_, T_thresh, binarize)
    if _all_equal(num_rays):
        return (weights, weights_sum_.reshape(num_scenes, num_rays[0]), depth_.reshape(num_scenes, num_rays[0]),
                image_.reshape(num_scenes, num_rays[0], 3))
    return weights, weights_sum_.split(num_rays, dim=0), depth_.split(num_rays, dim=0), image_.split(num_rays, dim=0)


# ----------------------------------------------------------------------------
# infer
# ----------------------------------------------------------------------------
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
               perturb=False, dt_gamma=0, max_steps=1024, contract=False, noises=None):
    """-> xyzs [n_alive*n_step,3], dirs [..,3], ts [..,2]  (raymarching.py:436-488)."""
    rays_o = _gpu_f32(rays_o).view(-1, 3)
    rays_d = _gpu_f32(rays_d).view(-1, 3)
    dev = rays_o.device
    M = n_alive * n_step
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    ts = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    if noises is not None:
        noises = _gpu_f32(noises)
    elif perturb:
        noises = torch.rand(n_alive, dtype=torch.float32, device=dev)
    else:
        noises = torch.zeros(n_alive, dtype=torch.float32, device=dev)
    assert rays_alive.is_cuda and rays_alive.dtype == torch.int32 and rays_alive.is_contiguous()
    assert rays_t.is_cuda and rays_t.dtype == torch.float32 and rays_t.is_contiguous()
    near = _gpu_f32(near)
    far = _gpu_f32(far)
    density_bitfield = _gpu(density_bitfield).contiguous()
    with torch.cuda.device(dev):
        _lib.call('mve_march_rays', int(n_alive), int(n_step), _lib.ptr(rays_alive), _lib.ptr(rays_t), _lib.ptr(rays_o),
                  _lib.ptr(rays_d), float(bound), int(bool(contract)), float(dt_gamma), int(max_steps), int(C), int(H),
                  _lib.ptr(density_bitfield), _lib.ptr(near), _lib.ptr(far), _lib.ptr(xyzs), _lib.ptr(dirs),
                  _lib.ptr(ts), _lib.ptr(noises), _s(rays_o))
    return xyzs, dirs, ts


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh=1e-2,
                   binarize=False):
    """In place on rays_alive / rays_t / weights_sum / depth / image  (raymarching.py:494-524)."""
    sigmas = _gpu_f32(sigmas)
    rgbs = _gpu_f32(rgbs)
    for t in (rays_t, ts, weights_sum, depth, image):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert rays_alive.is_cuda and rays_alive.dtype == torch.int32 and rays_alive.is_contiguous()
    with torch.cuda.device(sigmas.device):
        _lib.call('mve_composite_rays', int(n_alive), int(n_step), float(T_thresh), int(bool(binarize)),
                  _lib.ptr(rays_alive), _lib.ptr(rays_t), _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(ts),
                  _lib.ptr(weights_sum), _lib.ptr(depth), _lib.ptr(image), _s(sigmas))
    return tuple()


def compact_alive(rays_alive, n_alive):
    """Order-preserving device compaction of rays_alive[:n_alive] >= 0; replaces the boolean-mask
    gather of base_volume_renderer.py:322.  Returns (compacted tensor [n_alive], n_kept device int32[1])."""
    dev = rays_alive.device
    out = torch.empty(max(int(n_alive), 1), dtype=torch.int32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(_lib.raw('mve_march_scratch_bytes')(int(n_alive)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.call('mve_compact_alive', _lib.ptr(rays_alive), int(n_alive), _lib.ptr(out), _lib.ptr(n_out),
                  _lib.ptr(scratch), _s(rays_alive))
    return out, n_out


def cull_samples(weights, threshold, xyzs, dirs, ts, rays):
    """Train-branch weight culling (base_volume_renderer.py:222-243): keep samples with weight > threshold (order preserving)
    and re-index rays.  -> xyzs', dirs', ts', rays' (new tensors)."""
    weights = _gpu_f32(weights)
    dev = weights.device
    M, N = weights.shape[0], rays.shape[0]
    o_xyzs, o_dirs, o_ts = torch.empty_like(xyzs), torch.empty_like(dirs), torch.empty_like(ts)
    o_rays = torch.empty_like(rays)
    pref = torch.empty(M + 1, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(_lib.raw('mve_cull_scratch_bytes')(M), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.call('mve_cull_samples', _lib.ptr(weights), M, float(threshold), _lib.ptr(rays), N, _lib.ptr(xyzs), _lib.ptr(dirs),
                  _lib.ptr(ts), _lib.ptr(o_xyzs), _lib.ptr(o_dirs), _lib.ptr(o_ts), _lib.ptr(o_rays), _lib.ptr(pref), _lib.ptr(total),
                  _lib.ptr(scratch), _s(weights))
    k = int(total.item())      # the reference's boolean-mask indexing synchronises here as well
    return o_xyzs[:k], o_dirs[:k], o_ts[:k], o_rays
