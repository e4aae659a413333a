"""CPU restatement of the texture back-projection `MeshRenderer.bake_multiview`
(/root/reference/lib/models/decoders/mesh_renderer/base_mesh_renderer.py:507-603).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's rasterisation, interpolation and texture fetches are nvdiffrast calls (requirements.txt:3,
not available here).  This file restates the reference's data flow line by line on top of oracle/raster.py, with
  * dr.texture(..., filter_mode) restated as a plain bilinear fetch with wrap addressing (nvdiffrast's default boundary
    mode) for texture_filter='linear', and as the mip-mapped trilinear fetch of oracle/texture_mip_oracle.py for the reference's
    default 'linear-mipmap-linear';
  * the visibility term d sum(dr.texture(ones, texc)) / d ones  (:547-552) restated as what that gradient is for the
    bilinear filter: the scatter-add of each foreground pixel's four bilinear weights.
The geometry helpers it calls (get_ray_directions, depth_to_normal) ARE pinned against reference-executed golden vectors
(tests/golden/reference_py.npz).
"""
import numpy as np

from . import nerf_oracle as NO
from . import raster as OR

f32 = np.float32


def project(v, poses, intrinsics, h, w, near=0.1, far=10.0):
    """base_mesh_renderer.py:527-541 -> (v_cam [n,V,3], v_clip [n,V,4]) float32."""
    poses = np.asarray(poses, f32)
    intr = np.asarray(intrinsics, f32)
    n = poses.shape[0]
    r = np.concatenate([poses[:, :3, :1], -poses[:, :3, 1:3]], axis=-1)
    proj = np.zeros((n, 4, 4), f32)
    proj[:, 0, 0] = 2 * intr[:, 0] / w
    proj[:, 0, 2] = -2 * intr[:, 2] / w + 1
    proj[:, 1, 1] = -2 * intr[:, 1] / h
    proj[:, 1, 2] = -2 * intr[:, 3] / h + 1
    proj[:, 2, 2] = -(far + near) / (far - near)
    proj[:, 2, 3] = -(2 * far * near) / (far - near)
    proj[:, 3, 2] = -1
    v_cam = np.einsum('nvk,nkj->nvj', (v[None] - poses[:, None, :3, 3]).astype(f32), r).astype(f32)
    v_h = np.concatenate([v_cam, np.ones_like(v_cam[..., :1])], axis=-1)
    v_clip = np.einsum('nvk,njk->nvj', v_h, proj).astype(f32)
    return v_cam, v_clip


def _taps(u, v, nx, ny):
    """bilinear taps of uv on an nx x ny texel grid, centres at (i+0.5)/n, wrap addressing; all float32."""
    x = u.astype(f32) * f32(nx) - f32(0.5)
    y = v.astype(f32) * f32(ny) - f32(0.5)
    fx, fy = np.floor(x), np.floor(y)
    wx1 = (x - fx).astype(f32); wx0 = (f32(1) - wx1).astype(f32)
    wy1 = (y - fy).astype(f32); wy0 = (f32(1) - wy1).astype(f32)
    ix0 = np.mod(fx.astype(np.int64), nx); ix1 = np.mod(fx.astype(np.int64) + 1, nx)
    iy0 = np.mod(fy.astype(np.int64), ny); iy1 = np.mod(fy.astype(np.int64) + 1, ny)
    return (ix0, ix1), (iy0, iy1), (wx0, wx1), (wy0, wy1)


def texture_bilinear(tex, uv):
    """tex [h,w,c], uv [...,2] -> [...,c]; accumulation order (j outer, k inner) as csrc/raster.hip."""
    h, w, _ = tex.shape
    ix, iy, wx, wy = _taps(uv[..., 0], uv[..., 1], w, h)
    out = np.zeros(uv.shape[:-1] + (tex.shape[-1],), f32)
    for j in range(2):
        for k in range(2):
            out = (out + (wx[k] * wy[j]).astype(f32)[..., None] * tex[iy[j], ix[k]]).astype(f32)
    return out


def texture_mip(tex, uv, pos, tri, rast, attr, attr_tri, tex_attr_batch=None):
    """dr.texture(tex [Bt,H,W,C], uv [n,h,w,2], uv_da, 'linear-mipmap-linear') with uv_da = the pixel differentials of `attr` (the
    attribute uv was interpolated from: [V,2] or [n,V,2]) over the rasterisation (pos [n,V,4], tri, rast) -- numpy in, numpy out, through
    oracle/texture_mip_oracle.py."""
    import torch
    from . import texture_mip_oracle as T
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    rast_t = t(rast.astype(f32))
    db = T.rasterize_db(t(pos.astype(f32)), t(np.asarray(tri)), rast_t)
    da = T.interpolate_da(t(np.asarray(attr, f32)), rast_t, db, t(np.asarray(attr_tri)))
    return T.texture(t(tex.astype(f32)), t(uv.astype(f32)), da).numpy()


def visibility_mip(texc, pos, tri, rast, vt, ft, map_size):
    """`visibility_grad` with the mip-mapped filter (:466-475, :547-552): gradient of sum(dr.texture(ones, texc, uv_da)) w.r.t. the ones,
    over the covered pixels (the background pixels' uv = 0 splat onto the atlas corner is left out, as in splat_visibility)."""
    import torch
    from . import texture_mip_oracle as T
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    rast_t = t(rast.astype(f32))
    db = T.rasterize_db(t(pos.astype(f32)), t(np.asarray(tri)), rast_t)
    da = T.interpolate_da(t(np.asarray(vt, f32)), rast_t, db, t(np.asarray(ft)))
    n = rast.shape[0]
    ones = torch.ones(n, map_size, map_size, 1, dtype=torch.float64, requires_grad=True)
    out = T.texture(ones, t(texc.astype(f32)).double(), da.double())
    fg = (rast_t[..., 3] > 0)[..., None]
    g, = torch.autograd.grad((out * fg).sum(), ones)
    return g[..., 0].numpy()


def splat_visibility(texc, fg, map_size):
    """sum of bilinear footprint weights per texel; texc [h,w,2], fg [h,w] -> [map,map] float64."""
    ix, iy, wx, wy = _taps(texc[fg][:, 0], texc[fg][:, 1], map_size, map_size)
    vis = np.zeros((map_size, map_size), np.float64)
    for j in range(2):
        for k in range(2):
            np.add.at(vis, (iy[j], ix[k]), (wx[k] * wy[j]).astype(f32).astype(np.float64))
    return vis


def view_weight(depth, alpha, intrinsics, cos_weight_pow):
    """:559-566: cos weight from depth normals, times alpha, 5x5 min-pool.  depth [n,h,w], alpha [n,h,w] -> [n,h,w]."""
    n, h, w = depth.shape
    dirs = NO.get_ray_directions(h, w, intrinsics)
    dirs = (dirs / np.maximum(np.linalg.norm(dirs, axis=-1, keepdims=True), f32(1e-12))).astype(f32)      # norm=True
    normals = NO.depth_to_normal(depth, dirs, format='opencv') * f32(2) - f32(1)
    cosw = np.maximum(-(normals * dirs).sum(-1), 0).astype(f32)
    wimg = (np.power(cosw, f32(cos_weight_pow)) * alpha).astype(f32)
    pad = np.pad(wimg, ((0, 0), (2, 2), (2, 2)), constant_values=np.inf)
    out = np.full_like(wimg, np.inf)
    for dy in range(5):
        for dx in range(5):
            out = np.minimum(out, pad[:, dy:dy + h, dx:dx + w])
    return out, wimg


def bake_multiview(v, f, vt, ft, images, alphas, poses, intrinsics, map_size, cos_weight_pow=8.0, near=0.1, far=10.0,
                   projected=None, dilate=None, texture_filter='linear'):
    """Returns (albedo [map,map,3] before dilation/clamp, accum [map,map,4], valid [map,map], per-view debug dict)."""
    n, h, w, _ = images.shape
    vt = np.asarray(vt, f32)
    vt_clip = np.concatenate([vt * 2 - 1, np.tile(np.array([[0., 1.]], f32), (vt.shape[0], 1))], axis=-1)[None]
    tex_rast = OR.rasterize(vt_clip, ft, (map_size, map_size))[0]
    valid = tex_rast[..., 3] > 0
    v_cam, v_clip = projected if projected is not None else project(v, poses, intrinsics, h, w, near, far)
    rast = OR.rasterize(v_clip, f, (h, w))
    texc = OR.interpolate(vt[None], rast, ft)
    fg = rast[..., 3] > 0
    with np.errstate(divide='ignore'):
        depth = (f32(1) / OR.interpolate(-v_cam[..., 2:3], rast, f)[..., 0]).astype(f32)
    depth[~fg] = 0
    wimg, cosw = view_weight(depth, alphas[..., 0], intrinsics, cos_weight_pow)
    v_img = (v_clip[..., :2] / v_clip[..., 3:] * f32(0.5) + f32(0.5)).astype(f32)
    accum = np.zeros((map_size, map_size, 4), f32)
    vis_all = []
    mip = texture_filter == 'linear-mipmap-linear'
    vis_mip = visibility_mip(texc, v_clip, f, rast, vt, ft, map_size) if mip else None
    for i in range(n):
        vis = vis_mip[i] if mip else splat_visibility(texc[i], fg[i], map_size)
        vis_all.append(vis)
        imgc = OR.interpolate(v_img[i:i + 1], tex_rast[None], f)[0]
        img4 = np.concatenate([images[i], wimg[i][..., None]], axis=-1).astype(f32)
        if mip:       # :573-577: image-space coordinates and their differentials over the ATLAS rasterisation
            tex = texture_mip(img4[None], imgc[None], vt_clip, ft, tex_rast[None], v_img[i], f)[0]
        else:
            tex = texture_bilinear(img4, imgc)
        weight = (tex[..., 3] * vis.astype(f32)).astype(f32)
        accum[..., :3] = accum[..., :3] + tex[..., :3] * weight[..., None]
        accum[..., 3] = accum[..., 3] + weight
    albedo = accum[..., :3] / np.maximum(accum[..., 3:], f32(1e-8))
    return albedo.astype(f32), accum, valid, dict(vis=np.stack(vis_all), wimg=wimg, depth=depth, rast=rast, tex_rast=tex_rast)
