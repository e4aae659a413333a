"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Restatement of `MeshRenderer.forward`, single-scene branch (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:207-395) on top of
the raster oracle: depth = 1 / interpolate(-z_cam), normals rotated into the camera frame and mapped to [0, 1], albedo from the texture
(bilinear, or mip-mapped with texture_filter='linear-mipmap-linear') or the vertex colours, optional edge dilation, antialias over the packed (rgba, depth, normal) image, SSAA box filter.

PINNED (orchestration): tests/test_mesh_forward_ref.py compares it with the reference's own `forward` executed over a stand-in `dr`
module (tests/golden/make_mesh_forward_golden.py).  The raster / antialias / filter primitives stay this repo's specification
(nvdiffrast is absent)."""
import numpy as np

from . import bake_oracle as BO
from . import raster as OR

f32 = np.float32


def mesh_forward(v, f, vn, fn, projected, r_c2w, h, w, vt=None, ft=None, albedo=None, vc=None, normal_bg=(0.5, 0.5, 1.0), aa=True, ssaa=1,
                 shading_fun=None, dilate=None, texture_filter='linear'):
    """projected = (v_cam [n,V,3], v_clip [n,V,4]) at the SSAA resolution; r_c2w [n,3,3] (OpenGL camera-to-world rotation).
    -> rgba [n,h,w,4], depth [n,h,w], normal [n,h,w,3] at the output resolution."""
    v_cam, v_clip = projected
    n = v_clip.shape[0]
    H, W = h * ssaa, w * ssaa
    rast = OR.rasterize(v_clip, f, (H, W))
    fg = rast[..., 3] > 0
    with np.errstate(divide='ignore'):
        depth = (f32(1) / OR.interpolate(-v_cam[..., 2:3], rast, f)[..., 0]).astype(f32)
    depth[~fg] = 0
    nrm = OR.interpolate(vn[None], rast, fn)
    nrm = (nrm / np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), f32(1e-12))).astype(f32)
    rot = (np.einsum('bhwk,bkj->bhwj', nrm, r_c2w) / f32(2) + f32(0.5)).astype(f32)
    rot[~fg] = np.asarray(normal_bg, f32)
    alpha = fg[..., None].astype(f32)
    if vt is not None and albedo is not None:
        texc = OR.interpolate(vt[None], rast, ft)
        if texture_filter == 'linear-mipmap-linear':          # :241, :260-264 (oracle/texture_mip_oracle.py)
            alb = BO.texture_mip(albedo[..., :3][None], texc, v_clip, f, rast, vt, ft)
        else:
            alb = np.stack([BO.texture_bilinear(albedo[..., :3], texc[i]) for i in range(n)])
        alb[~fg] = 0
    elif vc is not None:
        rgba_v = OR.interpolate(vc[None], rast, f)
        alpha = alpha * rgba_v[..., 3:4]
        alb = rgba_v[..., :3] * alpha
    else:
        alb = np.zeros_like(rot)
    if shading_fun is not None:
        xyz = OR.interpolate(v[None], rast, f)
        out = np.zeros_like(alb)
        out[fg] = shading_fun(world_pos=xyz[fg], albedo=alb[fg], world_normal=nrm[fg], fg_mask=fg)
        alb = out
    rgba = np.concatenate([alb, alpha], -1).astype(f32)
    if dilate is not None:
        rgba = dilate(rgba)
    if aa:
        packed = OR.antialias(np.concatenate([rgba, depth[..., None], rot], -1), rast, v_clip, f)
        rgba, depth, rot = packed[..., :4], packed[..., 4], packed[..., 5:]
    if ssaa > 1:
        box = lambda x: x.reshape(n, h, ssaa, w, ssaa, -1).mean(axis=(2, 4), dtype=np.float64).astype(f32)
        rgba, depth, rot = box(rgba), box(depth[..., None])[..., 0], box(rot)
    return rgba, depth, rot


def edge_dilation(img, mask, radius=3, iters=7):
    """lib/ops/edge_dilation.py:5-47 in numpy: img [n,c,h,w], mask [n,1,h,w] (0/1 floats).  Every iteration fills the empty pixels that
    have a valid pixel within the (2r+1)^2 window with the NEAREST valid one (ties: first in row-major window order, as torch.argmax).
    Pinned bit-exactly against the reference's output (tests/golden/reference_py.npz, tests/test_mesh_forward_ref.py)."""
    if radius == 0 or iters == 0:
        return img
    img, mask = img.astype(f32).copy(), mask.astype(f32).copy()
    n, c, h, w = img.shape
    r = int(round(radius))
    k = 2 * r + 1
    d1 = np.linspace(-r, r, k, dtype=f32) ** 2
    dist = np.sqrt(d1[None, :] + d1[:, None]).astype(f32)
    score = (dist.max() - dist + 1).astype(f32).reshape(-1)
    for _ in range(iters):
        pad = np.pad(mask[:, 0], ((0, 0), (r, r), (r, r)))
        unfold = np.stack([pad[:, dy:dy + h, dx:dx + w] for dy in range(k) for dx in range(k)], -1)      # [n,h,w,k*k], F.unfold order
        mask_out = unfold.max(-1)
        do_fill = (mask_out - mask[:, 0]) > 0.5
        ind = (unfold * score).argmax(-1)
        bn, yy, xx = np.nonzero(do_fill)
        sel = ind[bn, yy, xx]
        sy, sx = yy + sel // k - r, xx + sel % k - r
        out = img.copy()
        out[bn, :, yy, xx] = img[bn, :, sy, sx]
        img, mask = out, mask_out[:, None]
    return img
