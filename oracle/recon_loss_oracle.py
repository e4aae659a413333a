"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain torch, of the image-space part of one NeRF optimisation iteration of the reference:
lib/pipelines/mvedit_3d_pipeline.py:542-603 (`nerf_optim`, from `out_rgbs = outputs['image']...` to `loss = loss + entropy_loss`) and
the in-tree functions it calls -- `depth_to_normal` (lib/core/utils/geometry_utils.py:119-148), `tv_loss` (lib/models/losses/tv_loss.py:
8-40, power 1.5, dims (-2, -1)), `l1_loss_mod` (lib/models/losses/pixelwise_loss.py:10-21) under mmgen's `weighted_loss` (element-wise
loss * weight, mean), `Tonemapping.lut / inverse_lut` (lib/models/decoders/tonemapping.py:33-53).

PINNED: tests/golden/recon_loss_ref.npz holds the values and autograd gradients of the reference's OWN statements executed on the CPU in
float64 (tests/golden/make_recon_loss_golden.py); tests/test_recon_loss.py checks this restatement against them to 1e-10.
Differentiable (torch autograd), any float dtype."""
import math

import torch
import torch.nn.functional as F


def depth_to_normal(depth, directions):
    """geometry_utils.py:119-148, format='opengl'"""
    xyz = directions / depth.unsqueeze(-1).clamp(min=1e-6)
    dx = xyz[..., :, 1:, :] - xyz[..., :, :-1, :]
    dy = xyz[..., 1:, :, :] - xyz[..., :-1, :, :]
    right = torch.cat([dx, dx[..., :, -1:, :]], dim=-2)
    left = torch.cat([-dx[..., :, :1, :], -dx], dim=-2)
    up = torch.cat([-dy[..., :1, :, :], -dy], dim=-3)
    down = torch.cat([dy, dy[..., -1:, :, :]], dim=-3)

    def nrm(v):
        return v / v.norm(dim=-1, keepdim=True).clamp(min=1e-12)
    n = nrm(nrm(torch.cross(right, up, dim=-1)) + nrm(torch.cross(up, left, dim=-1)) + nrm(torch.cross(left, down, dim=-1))
            + nrm(torch.cross(down, right, dim=-1)))
    n = torch.cat([n[..., :1], -n[..., 1:3]], dim=-1)
    return n / 2 + 0.5


def tv_loss(pred, target, weight, power=1.5):
    """pred / target / weight [P, C, h, w] (weight [P, 1, h, w]); tv_loss.py:8-40 followed by the mean of weighted_loss"""
    def diffs(t):
        dh = torch.cat([t[:, :, 1:] - t[:, :, :-1], torch.zeros_like(t[:, :, :1])], dim=2)
        dw = torch.cat([t[:, :, :, 1:] - t[:, :, :, :-1], torch.zeros_like(t[:, :, :, :1])], dim=3)
        return dh, dw
    dh, dw = diffs(pred)
    if target is not None:
        th, tw = diffs(target)
        dh, dw = dh - th, dw - tw
    wh = torch.cat([torch.minimum(weight[:, :, :-1], weight[:, :, 1:]), torch.zeros_like(weight[:, :, :1])], dim=2)
    ww = torch.cat([torch.minimum(weight[:, :, :, :-1], weight[:, :, :, 1:]), torch.zeros_like(weight[:, :, :, :1])], dim=3)
    return torch.stack([dh * wh, dw * ww], dim=0).norm(dim=0).pow(power).mean()


def _interp(a, b, v):
    i = torch.bucketize(v, a, right=True).clamp(min=1, max=len(a) - 1)
    t = (v - a[i - 1]) / (a[i] - a[i - 1])
    return b[i - 1] + (b[i] - b[i - 1]) * t


def nerf_optim_loss(image, weights_sum, depth, weights, bin_width, target_rgbs, target_m_blur, target_dir, patch_w, patch_lights, *,
                    target_n=None, target_depth=None, lut_x=None, lut_y=None, shaded=True, is_init=False, ambient_light=0.2, bg_color=1.0,
                    normal_bg=(0.5, 0.5, 1.0), pixel_loss_weight=1.2, normal_reg_weight=0.0, depth_weight=0.0, entropy_weight=0.0,
                    bg_width=0.015):
    """image [N, 3], weights_sum [N], depth [N] (N = P * ps * ps rays, patch-major), weights / bin_width [M] per sample;
    target_* [P, ps, ps, C]; patch_w [P] (= cam_weights[cam_ids] / cam_weights_mean), patch_lights [P, 3].
    `shaded` = `not is_init or init_shaded` (:558).  Returns a dict with `loss`, its parts, `out_rgbs`, `out_normals`."""
    P, ps = target_rgbs.shape[:2]
    dt = image.dtype
    nbg = torch.as_tensor(normal_bg, dtype=dt, device=image.device)
    out_alphas = weights_sum.reshape(P, ps, ps, 1)
    out_depth = depth.reshape(P, ps, ps) * torch.linalg.norm(target_dir, dim=-1)
    out_depth_fg = out_depth / out_alphas[..., 0].clamp(min=1e-6)
    nfg = depth_to_normal(out_depth_fg, target_dir)
    out_normals = nfg * out_alphas + nbg * (1 - out_alphas)
    wfg = -F.max_pool2d(-out_alphas.detach().squeeze(-1).unsqueeze(1), 3, stride=1, padding=1)     # [P, 1, ps, ps]
    out_rgbs = image.reshape(P, ps, ps, 3)
    w = patch_w[:, None, None, None]
    if shaded:
        ncv = torch.cat([nfg[..., :1] * 2 - 1, -nfg[..., 1:3] * 2 + 1], dim=-1)
        shading = ((patch_lights[:, None, None, :] * ncv).sum(-1, keepdim=True).clamp(min=0) * (1 - ambient_light) + ambient_light)
        if lut_x is None:
            out_rgbs = out_rgbs * shading + bg_color * (1 - out_alphas)
        else:
            out_rgbs = _interp(lut_x, lut_y, _interp(lut_y, lut_x, out_rgbs / out_alphas.clamp(min=1e-6))
                               + shading.clamp(min=1e-6).log2()) * out_alphas + bg_color * (1 - out_alphas)
    else:
        out_rgbs = out_rgbs + bg_color * (1 - out_alphas)
    res = dict(out_rgbs=out_rgbs, out_normals=out_normals, out_normals_fg=nfg, out_normals_fg_weight=wfg)
    res['pixel_rgb_loss'] = ((out_rgbs - target_rgbs).abs() * w).mean() * pixel_loss_weight * 4.5
    res['alphas_loss'] = ((out_alphas - target_m_blur).abs() * w).mean() * pixel_loss_weight * (5.0 if is_init else 1.0)
    res['normal_reg_loss'] = tv_loss(nfg.permute(0, 3, 1, 2), None if target_n is None else target_n.permute(0, 3, 1, 2), wfg) \
        * (normal_reg_weight * 10)
    loss = res['pixel_rgb_loss'] + res['alphas_loss'] + res['normal_reg_loss']
    if target_depth is not None:
        res['depth_loss'] = ((out_depth.reshape(target_depth.shape) - target_depth).abs() * w).mean() * pixel_loss_weight * depth_weight
        loss = loss + res['depth_loss']
    weights, bin_width = weights.float(), bin_width.float()        # :596-597 (a no-op on the fp32 tensors of a real run)
    bg_w = 1 - weights_sum.flatten()
    res['entropy_loss'] = -(torch.sum(weights * (torch.log(weights.clamp(min=1e-6)) - torch.log(bin_width.clamp(min=1e-6))))
                            + torch.sum(bg_w * (torch.log(bg_w.clamp(min=1e-6)) - math.log(bg_width)))) * (entropy_weight / (P * ps * ps))
    res['loss'] = loss + res['entropy_loss']
    return res


def mesh_optim_loss(rgba, normal, depth, target_rgbs, target_m_erode, target_m_blur, target_dir, view_w, *, target_n=None, simplified=False,
                    normal_bg=(0.5, 0.5, 1.0), pixel_loss_weight=1.2, normal_reg_weight=0.0):
    """Image-space part of one MESH optimisation iteration, lib/pipelines/mvedit_3d_pipeline.py:745-782 (`mesh_optim`): rgba [n, S, S, 4],
    normal [n, S, S, 3], depth [n, S, S] as `render_out` holds them; view_w [n] = cam_weights / cam_weights_mean; simplified =
    `mesh_is_simplified`.  PINNED by tests/golden/mesh_loss_ref.npz (the reference's own statements executed, float64)."""
    dt = rgba.dtype
    nbg = torch.as_tensor(normal_bg, dtype=dt, device=rgba.device)
    w = view_w[:, None, None, None]
    out_alphas = rgba[..., 3:]
    out_rgbs = rgba[..., :3] / out_alphas.clamp(min=1e-3)
    out_rgbs = out_rgbs * target_m_erode + target_rgbs * (1 - target_m_erode)
    n_cv = depth_to_normal_opencv(depth.detach(), target_dir) * 2 - 1
    cos = (n_cv * F.normalize(target_dir, dim=-1)).sum(-1, keepdim=True).neg().clamp(min=0)
    cos = -F.max_pool2d(-cos.permute(0, 3, 1, 2), 5, stride=1, padding=2).permute(0, 2, 3, 1)
    out_normals = normal * cos + normal.detach() * (1 - cos)
    nfg = (out_normals - nbg * (1 - out_alphas)) / out_alphas.clamp(min=1e-3)
    res = dict(out_rgbs=out_rgbs, out_normals=out_normals, out_normals_cos=cos)
    res['pixel_rgb_loss'] = ((out_rgbs - target_rgbs).abs() * w).mean() * pixel_loss_weight * 4.5
    loss = res['pixel_rgb_loss']
    if not simplified:
        res['alphas_loss'] = ((out_alphas - target_m_blur).abs() * w).mean() * pixel_loss_weight * 2.0
        res['normal_reg_loss'] = tv_loss(nfg.permute(0, 3, 1, 2), None if target_n is None else target_n.permute(0, 3, 1, 2),
                                         out_alphas.detach().permute(0, 3, 1, 2)) * (normal_reg_weight * 2)
        loss = loss + res['alphas_loss'] + res['normal_reg_loss']
    res['loss'] = loss
    return res


def depth_to_normal_opencv(depth, directions):
    """geometry_utils.py:119-148, format='opencv': depth_to_normal without the axis flips"""
    n = depth_to_normal(depth, directions) * 2 - 1
    n = torch.cat([n[..., :1], -n[..., 1:3]], dim=-1)
    return n / 2 + 0.5
