"""ORACLE -- test infrastructure only (imported by tests/, tests/golden/make_*.py; never by the product).

nvdiffrast's mip-mapped texture path as the reference uses it everywhere (lib/models/decoders/mesh_renderer/base_mesh_renderer.py:196
`texture_filter='linear-mipmap-linear'`; :241 / :442 / :466 `rast, rast_db = dr.rasterize(...)`; :260-264, :357-361, :467-474, :496-500,
:544-551, :573-577 `dr.interpolate(..., rast_db=rast_db, diff_attrs='all')` -> `dr.texture(tex, uv, uv_da=..., filter_mode=...)`).

nvdiffrast (requirements.txt: git+https://github.com/NVlabs/nvdiffrast.git@c5caf7bdb8a2448acc491a9faa47753972edd380) is NOT in /root/reference:
PARITY UNPINNED.  What follows restates its published algorithm (nvdiffrast/common/rasterize.cu `RasterizeCudaFwdShaderKernel`,
common/interpolate.cu `InterpolateFwdKernelDa`, common/texture.cu `MipBuildKernel`, `calculateMipLevel`, `indexTextureLinear`,
`TextureFwdKernelLinearMipmapLinear`):

  rast_db   (du/dX, du/dY, dv/dX, dv/dY) of the perspective-correct barycentrics (u, v) = (b0, b1) w.r.t. the pixel coordinates, from the
            clip-space triangle: with p_i' = (x_i - f_x w_i, y_i - f_y w_i) at the pixel's NDC position f, a0 = p1' x p2', a1 = p2' x p0',
            a2 = p0' x p1', iw = 1 / (a0 + a1 + a2):  du/dX = (2 / W) iw (b0 (da0 + da1 + da2)/dx - da0/dx) ... with
            da0/dx = y2 w1 - y1 w2 etc.  (b0, b1 are the clamped values the rasteriser stores)
  attr_da   d attr / dX = du/dX (a0 - a2) + dv/dX (a1 - a2), d attr / dY likewise; channel order (dA0/dX, dA0/dY, dA1/dX, dA1/dY, ...)
  mip stack level l+1 = 2x2 box average of level l (a dimension of size 1 stays 1), down to 1x1
  level     s = u W_tex, t = v H_tex;  A = s_X^2 + t_X^2, B = s_Y^2 + t_Y^2, C = s_X s_Y + t_X t_Y;
            major^2 = (A + B)/2 + sqrt((A - B)^2/4 + C^2);  level = clamp(log2(major^2) / 2, 0, max_level);
            level0 = floor(level), level1 = min(level0 + 1, max), f = level - level0; magnification (level == 0) reads level 0 only
  fetch     per level: wrap u, v to [0, 1), u W_l - 1/2, bilinear over the 4 texels with wrapped indices; out = a + f (b - a)

All functions are torch and differentiable w.r.t. the texture (autograd through the gathers), which is what the stand-in `dr` module of the
golden generators needs: `visibility_grad = autograd.grad(dr.texture(ones, ...).sum(), ones)` (base_mesh_renderer.py:470-475)."""
import math

import torch


def rasterize_db(pos, tri, rast):
    """pos [B,V,4] clip space, tri [F,3] int, rast [B,H,W,4] (u, v, z/w, id+1) -> rast_db [B,H,W,4]; zeros on empty pixels."""
    B, H, W, _ = rast.shape
    dt = pos.dtype
    idx = rast[..., 3].long() - 1
    fg = idx >= 0
    t = tri.long()[idx.clamp(min=0)]                                           # [B,H,W,3]
    bi = torch.arange(B, device=pos.device)[:, None, None, None].expand(-1, H, W, 3)
    p = pos[bi, t]                                                             # [B,H,W,3,4]
    xs, ys, xo, yo = 2.0 / W, 2.0 / H, 1.0 / W - 1.0, 1.0 / H - 1.0
    px = torch.arange(W, device=pos.device, dtype=dt)[None, None, :].expand(B, H, W)
    py = torch.arange(H, device=pos.device, dtype=dt)[None, :, None].expand(B, H, W)
    fx, fy = xs * px + xo, ys * py + yo
    x, y, w = p[..., 0], p[..., 1], p[..., 3]
    xp, yp = x - fx[..., None] * w, y - fy[..., None] * w
    a0 = xp[..., 1] * yp[..., 2] - yp[..., 1] * xp[..., 2]
    a1 = xp[..., 2] * yp[..., 0] - yp[..., 2] * xp[..., 0]
    a2 = xp[..., 0] * yp[..., 1] - yp[..., 0] * xp[..., 1]
    iw = 1.0 / (a0 + a1 + a2)
    b0, b1 = rast[..., 0].to(dt), rast[..., 1].to(dt)
    dfxdx, dfydy = xs * iw, ys * iw
    da0dx = y[..., 2] * w[..., 1] - y[..., 1] * w[..., 2]
    da0dy = x[..., 1] * w[..., 2] - x[..., 2] * w[..., 1]
    da1dx = y[..., 0] * w[..., 2] - y[..., 2] * w[..., 0]
    da1dy = x[..., 2] * w[..., 0] - x[..., 0] * w[..., 2]
    da2dx = y[..., 1] * w[..., 0] - y[..., 0] * w[..., 1]
    da2dy = x[..., 0] * w[..., 1] - x[..., 1] * w[..., 0]
    datdx, datdy = da0dx + da1dx + da2dx, da0dy + da1dy + da2dy
    db = torch.stack([dfxdx * (b0 * datdx - da0dx), dfydy * (b0 * datdy - da0dy),
                      dfxdx * (b1 * datdx - da1dx), dfydy * (b1 * datdy - da1dy)], dim=-1)
    return torch.where(fg[..., None], db, torch.zeros_like(db))


def interpolate_da(attr, rast, rast_db, tri):
    """attr [V,C] or [B,V,C] -> attribute pixel differentials [B,H,W,2C] (dA0/dX, dA0/dY, dA1/dX, ...); zeros on empty pixels."""
    B, H, W, _ = rast.shape
    if attr.dim() == 2:
        attr = attr[None]
    if attr.shape[0] == 1:
        attr = attr.expand(B, -1, -1)
    idx = rast[..., 3].long() - 1
    fg = idx >= 0
    t = tri.long()[idx.clamp(min=0)]
    bi = torch.arange(B, device=attr.device)[:, None, None, None].expand(-1, H, W, 3)
    a = attr[bi, t]                                                            # [B,H,W,3,C]
    dsdu, dsdv = a[..., 0, :] - a[..., 2, :], a[..., 1, :] - a[..., 2, :]
    db = rast_db.to(attr.dtype)
    dx = db[..., 0:1] * dsdu + db[..., 2:3] * dsdv
    dy = db[..., 1:2] * dsdu + db[..., 3:4] * dsdv
    out = torch.stack([dx, dy], dim=-1).reshape(B, H, W, -1)
    return torch.where(fg[..., None], out, torch.zeros_like(out))


def build_mips(tex, max_level=None):
    """tex [Bt,H,W,C] -> list of levels (level 0 = tex itself)."""
    levels = [tex]
    h, w = tex.shape[1], tex.shape[2]
    while (h | w) > 1 and (max_level is None or len(levels) - 1 < max_level):
        assert (h == 1 or h % 2 == 0) and (w == 1 or w % 2 == 0), 'mip construction needs even extents at every level (nvdiffrast raises too)'
        t = levels[-1]
        if h > 1:
            t = 0.5 * (t[:, 0::2] + t[:, 1::2])
        if w > 1:
            t = 0.5 * (t[:, :, 0::2] + t[:, :, 1::2])
        h, w = max(h >> 1, 1), max(w >> 1, 1)
        levels.append(t)
    return levels


def mip_level(uv_da, tex_h, tex_w, max_level):
    """-> (level0 long, level1 long, frac) per pixel"""
    dsdx, dsdy = uv_da[..., 0] * tex_w, uv_da[..., 1] * tex_w
    dtdx, dtdy = uv_da[..., 2] * tex_h, uv_da[..., 3] * tex_h
    A, Bq, C = dsdx * dsdx + dtdx * dtdx, dsdy * dsdy + dtdy * dtdy, dsdx * dsdy + dtdx * dtdy
    major = 0.5 * (A + Bq) + torch.sqrt(0.25 * (A - Bq) * (A - Bq) + C * C)
    lvl = 0.5 * torch.log2(major)                                              # -inf where the footprint is zero: clamped to 0
    lvl = torch.nan_to_num(lvl, nan=0.0, posinf=float(max_level), neginf=0.0).clamp(0.0, float(max_level))
    l0 = torch.floor(lvl).long()
    l1 = torch.where(lvl > 0, (l0 + 1).clamp(max=max_level), l0)
    return l0, l1, torch.where(lvl > 0, lvl - l0.to(lvl.dtype), torch.zeros_like(lvl))


def _bilinear_wrap(level, bsel, uv):
    """level [Bt,h,w,C], bsel [N] long (texture of each sample), uv [N,2] -> [N,C]"""
    h, w = level.shape[1], level.shape[2]
    u = uv[:, 0] - torch.floor(uv[:, 0])
    v = uv[:, 1] - torch.floor(uv[:, 1])
    u, v = u * w - 0.5, v * h - 0.5
    iu0, iv0 = torch.floor(u).long(), torch.floor(v).long()
    fu, fv = (u - iu0.to(u.dtype))[:, None], (v - iv0.to(v.dtype))[:, None]
    iu1, iv1 = (iu0 + 1) % w, (iv0 + 1) % h
    iu0, iv0 = iu0 % w, iv0 % h
    a00, a10 = level[bsel, iv0, iu0], level[bsel, iv0, iu1]
    a01, a11 = level[bsel, iv1, iu0], level[bsel, iv1, iu1]
    top = a00 + fu * (a10 - a00)
    bot = a01 + fu * (a11 - a01)
    return top + fv * (bot - top)


def texture(tex, uv, uv_da=None, filter_mode='linear-mipmap-linear', max_mip_level=None, mips=None):
    """dr.texture(tex [Bt,H,W,C], uv [N,h,w,2], uv_da [N,h,w,4], filter_mode in {'linear', 'linear-mipmap-linear'}), boundary_mode='wrap'.
    Bt == 1 broadcasts over N."""
    N, h, w, _ = uv.shape
    C = tex.shape[-1]
    bsel = (torch.arange(N, device=uv.device) if tex.shape[0] == N else torch.zeros(N, dtype=torch.long, device=uv.device))
    bsel = bsel[:, None].expand(N, h * w).reshape(-1)
    uvf = uv.reshape(-1, 2)
    if filter_mode == 'linear' or uv_da is None:
        return _bilinear_wrap(tex, bsel, uvf).reshape(N, h, w, C)
    assert filter_mode == 'linear-mipmap-linear'
    levels = mips if mips is not None else build_mips(tex, max_mip_level)
    maxl = len(levels) - 1
    l0, l1, fr = mip_level(uv_da.reshape(-1, 4), tex.shape[1], tex.shape[2], maxl)
    out = torch.zeros(uvf.shape[0], C, dtype=tex.dtype, device=uv.device)
    for lv in range(maxl + 1):
        m0 = l0 == lv
        if m0.any():
            out[m0] = out[m0] + (1.0 - fr[m0])[:, None] * _bilinear_wrap(levels[lv], bsel[m0], uvf[m0])
        m1 = (l1 == lv) & (fr > 0)
        if m1.any():
            out[m1] = out[m1] + fr[m1][:, None] * _bilinear_wrap(levels[lv], bsel[m1], uvf[m1])
    return out.reshape(N, h, w, C)
