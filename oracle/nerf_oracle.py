"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (numpy / torch fp32) of the NeRF render path of the reference:
  * iNGPDecoder.point_decode            lib/models/decoders/ingp_decoder.py:106-120
  * VolumeRenderer.forward, eval branch  lib/models/decoders/base_volume_renderer.py:264-329
  * BaseNeRF.render                      lib/models/autoencoders/base_nerf.py:489-556
  * get_ray_directions / get_rays / depth_to_normal / normalize_depth
                                         lib/core/utils/geometry_utils.py:18-55, :119-168
The marcher / compositor calls go to oracle/raymarching.py (C restatement of raymarching.cu).

The hash-grid encoding is NOT in the reference tree: it is third-party `tinycudann` (requirements.txt:5, git
HEAD, unpinned), absent from this image.  It is restated from tiny-cuda-nn's published algorithm
(include/tiny-cuda-nn/encodings/grid.h and common_device.h as of 2023-2024):
    scale_l      = exp2f(l * log2f(per_level_scale)) * base_resolution - 1
    resolution_l = (uint32) ceilf(scale_l) + 1
    params_l     = min(next_multiple(resolution_l^3, 8), 2^log2_hashmap_size)      (offset table)
    pos          = fmaf(scale_l, x, 0.5);  cell = floor(pos);  frac = pos - cell
    w            = frac^2 (3 - 2 frac)                                               ("Smoothstep")
    index(cell)  = dense x + y R + z R^2 while the running stride fits, else
                   (x * 1) ^ (y * 2654435761) ^ (z * 805459861), finally  % params_l
    feature      = sum over the 8 corners, corner i uses (cell + bit_d(i)), weight prod_d (bit ? w_d : 1 - w_d)
PARITY UNPINNED for this piece (no tcnn here, version unpinned in the reference): self-consistency only.
The pure-torch geometry helpers ARE pinned: tests/golden/geometry_ref.npz is produced by executing the reference's
own function bodies (extracted from /root/reference with `ast`) in tests/golden/make_geometry_golden.py.
"""
import math

import numpy as np

from . import raymarching as ORM

PRIMES = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))


# ---------------------------------------------------------------------------------------------------
# hash grid (tiny-cuda-nn "HashGrid", Smoothstep) + MLP
# ---------------------------------------------------------------------------------------------------
def grid_meta(n_levels=12, base_resolution=16, max_resolution=320, bound=1.0, log2_hashmap_size=19):
    """-> list of (scale f32, resolution, offset, size) per level and total parameter rows."""
    pls = np.exp2(np.log2(max_resolution * bound / base_resolution) / (n_levels - 1))     # ingp_decoder.py:71
    log2_pls = np.float32(np.log2(np.float32(pls)))
    meta, off = [], 0
    for lvl in range(n_levels):
        scale = np.float32(np.exp2(np.float32(lvl) * log2_pls, dtype=np.float32) * np.float32(base_resolution) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        n = res ** 3
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        meta.append((scale, res, off, n))
        off += n
    return meta, off


def make_nerf_params(n_levels=12, max_resolution=320, hidden=64, seed=7, table_scale=1e-4):
    """Seeded random decoder state: hash table U(-1e-4,1e-4) (ingp_decoder.py:88), xavier-uniform MLP."""
    rng = np.random.default_rng(seed)
    meta, rows = grid_meta(n_levels, 16, max_resolution)
    table = rng.uniform(-table_scale, table_scale, (rows, 2)).astype(np.float32)
    d_in = 2 * n_levels

    def xavier(o, i):
        a = math.sqrt(6.0 / (i + o))
        return rng.uniform(-a, a, (o, i)).astype(np.float32)
    return dict(table=table, w1=xavier(hidden, d_in), b1=np.zeros(hidden, np.float32),
                w2=xavier(4, hidden), b2=np.zeros(4, np.float32), n_levels=n_levels, max_resolution=max_resolution)


def hashgrid_encode(x01, table, n_levels=12, max_resolution=320, bound=1.0, log2_hashmap_size=19):
    """x01: [M,3] float32 in [0,1] -> [M, 2*n_levels] float32."""
    x01 = np.ascontiguousarray(x01, dtype=np.float32)
    M = x01.shape[0]
    meta, _ = grid_meta(n_levels, 16, max_resolution, bound, log2_hashmap_size)
    out = np.zeros((M, 2 * n_levels), np.float32)
    for lvl, (scale, res, off, size) in enumerate(meta):
        pos = (scale * x01 + np.float32(0.5)).astype(np.float32)
        cell_f = np.floor(pos)
        frac = (pos - cell_f).astype(np.float32)
        cell = cell_f.astype(np.int64).astype(np.uint32)
        w = (frac * frac * (np.float32(3.0) - np.float32(2.0) * frac)).astype(np.float32)
        acc = np.zeros((M, 2), np.float32)
        for corner in range(8):
            weight = np.ones(M, np.float32)
            cg = np.empty((M, 3), np.uint32)
            for d in range(3):
                if corner & (1 << d):
                    weight = weight * w[:, d]
                    cg[:, d] = cell[:, d] + np.uint32(1)
                else:
                    weight = weight * (np.float32(1.0) - w[:, d])
                    cg[:, d] = cell[:, d]
            # grid_index
            stride = 1
            index = np.zeros(M, np.uint32)
            for d in range(3):
                if stride > size:
                    break
                index = index + cg[:, d] * np.uint32(stride & 0xFFFFFFFF)
                stride *= res
            if size < stride:
                index = (cg[:, 0] * PRIMES[0]) ^ (cg[:, 1] * PRIMES[1]) ^ (cg[:, 2] * PRIMES[2])
            index = index % np.uint32(size)
            acc = acc + weight[:, None] * table[off + index.astype(np.int64)]
        out[:, 2 * lvl:2 * lvl + 2] = acc
    return out


def point_decode(xyzs, params, bound=1.0, blob_density=1.0, blob_radius=0.2, sigmoid_saturation=0.001):
    """ingp_decoder.py:106-120 -> sigmas [M], rgbs [M,3] (float32)."""
    xyzs = np.ascontiguousarray(xyzs, dtype=np.float32)
    enc = hashgrid_encode((xyzs + np.float32(bound)) / np.float32(2 * bound), params['table'], params['n_levels'],
                          params['max_resolution'], bound)
    hdn = np.maximum(enc @ params['w1'].T + params['b1'], 0).astype(np.float32)
    h = (hdn @ params['w2'].T + params['b2']).astype(np.float32)
    d = np.maximum((xyzs * xyzs).sum(-1), np.float32(0.2))                               # density_blob, :100-103
    g = np.float32(blob_density) * np.exp(-d / np.float32(2 * blob_radius ** 2))
    sigmas = np.exp(h[:, 0] + g).astype(np.float32)                                      # TruncExp forward = exp
    rgbs = 1.0 / (1.0 + np.exp(-h[:, 1:].astype(np.float64)))
    rgbs = (rgbs * (1 + sigmoid_saturation * 2) - sigmoid_saturation).astype(np.float32)
    return sigmas, rgbs


# ---------------------------------------------------------------------------------------------------
# VolumeRenderer.forward, eval branch (base_volume_renderer.py:264-329): one scene
# ---------------------------------------------------------------------------------------------------
def render_rays_eval(rays_o, rays_d, bitfield, grid_size, params, bound=1.0, min_near=0.2, dt_gamma=0.0, max_steps=1024,
                     T_thresh=1e-2, decode=None):
    rays_o = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    N = rays_o.shape[0]
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = ORM.near_far_from_aabb(rays_o, rays_d, aabb, min_near)
    weights_sum, depth, image = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    rays_alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    decode = decode or (lambda x: point_decode(x, params, bound))
    step = 0
    n_samples = 0
    while step < max_steps:
        n_alive = rays_alive.shape[0]
        if n_alive == 0:
            break
        n_step = min(max(N // n_alive, 1), 8)
        xyzs, dirs, ts = ORM.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, 1, grid_size,
                                        nears, fars, np.zeros(n_alive, np.float32), dt_gamma, max_steps)
        n_samples += int((ts[:, 0] != 0).sum())
        sigmas, rgbs = decode(xyzs)
        ORM.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh)
        rays_alive = np.ascontiguousarray(rays_alive[rays_alive >= 0])
        step += n_step
    return weights_sum, depth, image, n_samples


# ---------------------------------------------------------------------------------------------------
# geometry helpers (geometry_utils.py) and BaseNeRF.render (base_nerf.py:489-556)
# ---------------------------------------------------------------------------------------------------
def get_ray_directions(h, w, intrinsics):
    """intrinsics [..., 4] (fx, fy, cx, cy) -> directions [..., h, w, 3] (z = 1), geometry_utils.py:18-39."""
    intr = np.asarray(intrinsics, np.float32)
    x = np.linspace(0.5, w - 0.5, w, dtype=np.float32)
    y = np.linspace(0.5, h - 0.5, h, dtype=np.float32)
    dx = (x - intr[..., 2:3]) / intr[..., 0:1]              # [..., w]
    dy = (y - intr[..., 3:4]) / intr[..., 1:2]              # [..., h]
    out = np.ones(intr.shape[:-1] + (h, w, 3), np.float32)
    out[..., 0] = dx[..., None, :]
    out[..., 1] = dy[..., :, None]
    return out


def get_rays(directions, c2w, norm=True):
    """geometry_utils.py:42-55: rays_d = directions @ R^T (normalised), rays_o = t."""
    c2w = np.asarray(c2w, np.float32)
    R = c2w[..., :3, :3]
    rays_d = np.einsum('...hwk,...jk->...hwj', directions, R).astype(np.float32)
    rays_o = np.broadcast_to(c2w[..., None, None, :3, 3], rays_d.shape).astype(np.float32)
    if norm:
        n = np.maximum(np.linalg.norm(rays_d, axis=-1, keepdims=True), np.float32(1e-12))      # F.normalize eps
        rays_d = (rays_d / n).astype(np.float32)
    return rays_o, rays_d


def _normalize(v):
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), np.float32(1e-12))


def depth_to_normal(depth, directions, format='opengl'):
    """geometry_utils.py:119-148; depth = 1/z."""
    xyz = directions / np.maximum(depth[..., None], np.float32(1e-6))
    dx = xyz[..., :, 1:, :] - xyz[..., :, :-1, :]
    dy = xyz[..., 1:, :, :] - xyz[..., :-1, :, :]
    right = np.concatenate([dx, dx[..., :, -1:, :]], axis=-2)
    left = np.concatenate([-dx[..., :, :1, :], -dx], axis=-2)
    up = np.concatenate([-dy[..., :1, :, :], -dy], axis=-3)
    down = np.concatenate([dy, dy[..., -1:, :, :]], axis=-3)
    n = _normalize(_normalize(np.cross(right, up)) + _normalize(np.cross(up, left)) + _normalize(np.cross(left, down))
                   + _normalize(np.cross(down, right)))
    n = n.astype(np.float32).copy()
    if format == 'opengl':
        n[..., 1:3] = -n[..., 1:3]
    else:
        assert format == 'opencv'
    return (n / 2 + 0.5).astype(np.float32)


def normalize_depth(depths, alphas, far_depth=0.25, alpha_clip=0.5, eps=1e-5):
    """geometry_utils.py:151-168.  depths [N,H,W], alphas [N,H,W,1]."""
    depths = np.asarray(depths, np.float32)
    a = np.asarray(alphas, np.float32)[..., 0]
    dmax = depths.reshape(depths.shape[0], -1).max(1)[:, None, None]
    fg = depths / np.maximum(a, np.float32(eps))
    fg_min = np.where(a < alpha_clip, np.float32(1 / eps), fg).reshape(depths.shape[0], -1).min(1)[:, None, None]
    fg = (fg - fg_min) / np.maximum(dmax - fg_min, np.float32(eps))
    fg = fg * np.float32(1 - far_depth) + np.float32(far_depth)
    return np.clip(fg * a, 0, 1).astype(np.float32)


def nerf_render(params, bitfield, grid_size, h, w, intrinsics, poses, dt_gamma_scale=0.0, normal_bg=(0.5, 0.5, 1.0),
                bound=1.0, min_near=0.2, max_steps=1024):
    """BaseNeRF.render with cfg=dict(return_rgba=True, compute_normal=True), one scene.
    intrinsics [b,4], poses [b,3,4] -> rgba [b,h,w,4], depth [b,h,w] (1/z), normal, normal_fg [b,h,w,3]."""
    intrinsics = np.asarray(intrinsics, np.float32)
    b = intrinsics.shape[0]
    dt_gamma = float(dt_gamma_scale * 2 / (intrinsics[:, 0] + intrinsics[:, 1]).mean())
    directions = get_ray_directions(h, w, intrinsics)
    rays_o, rays_d = get_rays(directions, poses, norm=True)
    ws, depth, image, _ = render_rays_eval(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), bitfield, grid_size, params, bound,
                                           min_near, dt_gamma, max_steps)
    rgba = np.concatenate([image, ws[:, None]], -1).reshape(b, h, w, 4)
    depth = depth.reshape(b, h, w) * np.linalg.norm(directions, axis=-1)                      # 1/r -> 1/z
    depth_fg = depth / np.maximum(rgba[..., 3], np.float32(1e-6))
    normal_fg = depth_to_normal(depth_fg, directions)
    normal = normal_fg * rgba[..., 3:] + np.asarray(normal_bg, np.float32) * (1 - rgba[..., 3:])
    return rgba.astype(np.float32), depth.astype(np.float32), normal.astype(np.float32), normal_fg


# ---------------------------------------------------------------------------------------------------
# train branch forward and density-grid refresh (lib/models/decoders/base_volume_renderer.py:105-262)
# ---------------------------------------------------------------------------------------------------
def cull_samples(weights, th, xyzs, dirs, ts, rays):
    """:222-243: boolean-mask gather + cumsum re-indexing of rays (offset, count)."""
    mask = weights > np.float32(th)
    filt = np.concatenate([[0], np.cumsum(mask)]).astype(np.int64)
    start = filt[rays[:, 0]]
    end = filt[rays[:, 0] + rays[:, 1]]
    return xyzs[mask], dirs[mask], ts[mask], np.stack([start, end - start], axis=-1).astype(np.int32), filt.astype(np.int32)


def density_grid_points(coords, noise, H, bound=1.0):
    """:129-135 / :158-160: Morton index and jittered centre of each cell."""
    coords = np.asarray(coords, np.int64)
    xyzs = ((coords.astype(np.float32) - np.float32((H - 1) / 2)) * np.float32(2 * bound / H)).astype(np.float32)
    hw = np.float32(bound / H)
    xyzs = (xyzs + (noise * (np.float32(2) * hw) - hw)).astype(np.float32)
    return xyzs, ORM.morton3D(coords.astype(np.int32))


def density_grid_update(grid, sigmas, indices, decay=0.9):
    """:140-141, :166-171 -> (new grid, mean of clamp(grid, 0))."""
    tmp = np.full_like(grid, -1)
    tmp[indices] = np.minimum(sigmas, np.finfo(np.float32).max)
    valid = (grid >= 0) & (tmp >= 0)
    out = np.where(valid, np.maximum(grid * np.float32(decay), tmp), grid).astype(np.float32)
    return out, np.float32(np.maximum(out, 0).astype(np.float64).mean())


def train_forward(rays_o, rays_d, bitfield, grid_size, params, noises, dt_gamma=0.0, bound=1.0, min_near=0.2, max_steps=1024,
                  weight_culling_th=1e-3):
    """VolumeRenderer.forward, self.training branch (:207-262) for one scene without normals."""
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = ORM.near_far_from_aabb(rays_o, rays_d, aabb, min_near)
    xyzs, dirs, ts, rays = ORM.march_rays_train(rays_o, rays_d, bound, bitfield, 1, grid_size, nears, fars, noises, dt_gamma, max_steps)
    if weight_culling_th > 0:
        sig, _ = point_decode(xyzs, params, bound)
        w, _, _, _ = ORM.composite_rays_train(sig, np.zeros((sig.shape[0], 3), np.float32), ts, rays)
        xyzs, dirs, ts, rays, _ = cull_samples(w, weight_culling_th, xyzs, dirs, ts, rays)
    sig, rgb = point_decode(xyzs, params, bound)
    weights, ws, depth, image = ORM.composite_rays_train(sig, rgb, ts, rays)
    return dict(weights=weights, weights_sum=ws, depth=depth, image=image, rays=rays, ts=ts, xyzs=xyzs)


# ---------------------------------------------------------------------------------------------------
# Decoder backward (the reconstruct step's gradients w.r.t. the hash table and the MLP).
# torch autograd over a torch restatement of point_decode above; the sigma activation's backward is the reference's own
# _trunc_exp (lib/ops/activation.py:8-20: grad * clamp(exp(x), 1e-6, 1e6)).
# ---------------------------------------------------------------------------------------------------
def decoder_grads_torch(xyzs, params, g_sigma, g_rgb, bound=1.0, blob_density=1.0, blob_radius=0.2, sigmoid_saturation=0.001):
    """-> dict(table, w1, b1, w2, b2) of float32 numpy gradients of  sum(g_sigma*sigma) + sum(g_rgb*rgb)."""
    import torch

    class TruncExp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            e = torch.exp(x)
            ctx.save_for_backward(e)
            return e

        @staticmethod
        def backward(ctx, g):
            return g * ctx.saved_tensors[0].clamp(min=1e-6, max=1e6)

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    x = t(np.asarray(xyzs, np.float32))
    table = t(params['table']).clone().requires_grad_(True)
    w1, b1 = t(params['w1']).clone().requires_grad_(True), t(params['b1']).clone().requires_grad_(True)
    w2, b2 = t(params['w2']).clone().requires_grad_(True), t(params['b2']).clone().requires_grad_(True)
    meta, _ = grid_meta(params['n_levels'], 16, params['max_resolution'], bound)
    u = (x + bound) / (2 * bound)
    feats = []
    M32 = 0xFFFFFFFF
    for (scale, res, off, size) in meta:
        pos = torch.tensor(scale, dtype=torch.float32) * u + 0.5
        cf = torch.floor(pos)
        fr = pos - cf
        w = fr * fr * (3.0 - 2.0 * fr)
        cell = cf.to(torch.int64) & M32
        acc = 0
        dense = res * res * res <= size and res <= size and res * res <= size
        for corner in range(8):
            wt = torch.ones(x.shape[0])
            c = []
            for d in range(3):
                if corner & (1 << d):
                    wt = wt * w[:, d]
                    c.append((cell[:, d] + 1) & M32)
                else:
                    wt = wt * (1.0 - w[:, d])
                    c.append(cell[:, d])
            if dense:
                idx = (c[0] + c[1] * res + c[2] * res * res) & M32
            else:
                idx = ((c[0] * 1) & M32) ^ ((c[1] * 2654435761) & M32) ^ ((c[2] * 805459861) & M32)
            idx = idx % size
            acc = acc + wt[:, None] * table[off + idx]
        feats.append(acc)
    enc = torch.cat(feats, dim=1)
    h = torch.relu(enc @ w1.t() + b1)
    o = h @ w2.t() + b2
    d2 = (x * x).sum(-1).clamp(min=0.2)
    sigma = TruncExp.apply(o[:, 0] + blob_density * torch.exp(-d2 / (2 * blob_radius ** 2)))
    rgb = torch.sigmoid(o[:, 1:]) * (1 + 2 * sigmoid_saturation) - sigmoid_saturation
    loss = (t(np.asarray(g_sigma, np.float32)) * sigma).sum() + (t(np.asarray(g_rgb, np.float32)) * rgb).sum()
    loss.backward()
    return dict(table=table.grad.numpy(), w1=w1.grad.numpy(), b1=b1.grad.numpy(), w2=w2.grad.numpy(), b2=b2.grad.numpy(),
                sigma=sigma.detach().numpy(), rgb=rgb.detach().numpy())
