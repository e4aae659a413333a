"""Host-side mirror of the image-space loss of one NeRF optimisation iteration (lib/pipelines/mvedit_3d_pipeline.py:542-603, `nerf_optim`
from `out_rgbs = outputs['image']...` to `loss = loss + entropy_loss`): native forward + backward (csrc/recon_loss.hip, five + three
launches, no atomics) behind one `torch.autograd.Function`, so the loop around it keeps calling `loss.backward()` and the gradients flow
on into `composite_rays_train`'s backward.  No PyTorch fallback.

    res = nerf_optim_loss(outputs['image'], outputs['weights_sum'], outputs['depth'], outputs['weights'], outputs['ts'][0],
                          target_rgbs, target_m_blur, target_dir, cam_weights[target_cam_ids] / cam_weights_mean, cam_lights[target_cam_ids],
                          target_n=..., target_depth=..., tonemapping=tm, shaded=not is_init or init_shaded, is_init=is_init, ...)
    loss = res['loss'] + patch_loss(res['out_rgbs'].permute(0, 3, 1, 2), ...) * patch_rgb_weight        # :611-616 unchanged
    loss.backward()"""
import ctypes

import torch

from . import _lib

PARTS = ('loss', 'pixel_rgb_loss', 'alphas_loss', 'normal_reg_loss', 'depth_loss', 'entropy_loss')


class _Desc(ctypes.Structure):
    """MveReconLossDesc (include/mvedit_amd.h)"""
    _fields_ = ([(n, ctypes.c_int32) for n in ('P', 'ps', 'shaded', 'is_init', 'lut_steps')]
                + [('ambient_light', ctypes.c_float), ('bg_color', ctypes.c_float), ('normal_bg', ctypes.c_float * 3)]
                + [(n, ctypes.c_float) for n in ('pixel_loss_weight', 'normal_reg_weight', 'depth_weight', 'entropy_weight', 'bg_width')]
                + [(n, ctypes.c_void_p) for n in ('d_lut_x', 'd_lut_y', 'd_image', 'd_weights_sum', 'd_depth', 'd_weights', 'd_ts')]
                + [('M', ctypes.c_uint32)]
                + [(n, ctypes.c_void_p) for n in ('d_target_dir', 'd_target_rgbs', 'd_target_m', 'd_target_n', 'd_target_depth', 'd_patch_w',
                                                  'd_patch_lights')])


def _f32(t, device):
    return None if t is None else t.detach().to(device=device, dtype=torch.float32).contiguous()


class _ReconLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, weights_sum, depth, weights, consts, hyper):
        dev = image.device
        assert image.is_cuda, 'native path: CUDA tensors only'
        P, ps = consts['target_rgbs'].shape[:2]
        N = P * ps * ps
        keep = dict(image=_f32(image, dev).reshape(N, 3), weights_sum=_f32(weights_sum, dev).reshape(N), depth=_f32(depth, dev).reshape(N),
                    weights=_f32(weights, dev).reshape(-1), **{k: _f32(v, dev) for k, v in consts.items()})
        M = keep['weights'].numel()
        assert keep['ts'].shape == (M, 2) and keep['target_dir'].numel() == 3 * N and keep['target_m'].numel() == N
        assert keep['patch_w'].numel() == P and keep['patch_lights'].shape == (P, 3)
        d = _Desc(P=P, ps=ps, shaded=int(hyper['shaded']), is_init=int(hyper['is_init']),
                  lut_steps=0 if keep['lut_x'] is None else keep['lut_x'].numel(), ambient_light=hyper['ambient_light'], bg_color=hyper['bg_color'],
                  normal_bg=(ctypes.c_float * 3)(*hyper['normal_bg']), pixel_loss_weight=hyper['pixel_loss_weight'],
                  normal_reg_weight=hyper['normal_reg_weight'], depth_weight=hyper['depth_weight'], entropy_weight=hyper['entropy_weight'],
                  bg_width=hyper['bg_width'], M=M)
        for field, key in (('d_lut_x', 'lut_x'), ('d_lut_y', 'lut_y'), ('d_image', 'image'), ('d_weights_sum', 'weights_sum'), ('d_depth', 'depth'),
                           ('d_weights', 'weights'), ('d_ts', 'ts'), ('d_target_dir', 'target_dir'), ('d_target_rgbs', 'target_rgbs'),
                           ('d_target_m', 'target_m'), ('d_target_n', 'target_n'), ('d_target_depth', 'target_depth'), ('d_patch_w', 'patch_w'),
                           ('d_patch_lights', 'patch_lights')):
            setattr(d, field, None if keep[key] is None else keep[key].data_ptr())
        ws_bytes = _lib.raw('mve_recon_loss_workspace_bytes')(P, ps, M)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)                  # owned by this call: backward reads it
        losses = torch.empty(6, dtype=torch.float32, device=dev)
        out_rgbs = torch.empty(P, ps, ps, 3, dtype=torch.float32, device=dev)
        out_normals = torch.empty(P, ps, ps, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call('mve_recon_loss_forward', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.ptr(losses), _lib.ptr(out_rgbs), _lib.ptr(out_normals),
                      _lib.stream_ptr(dev))
        # the descriptor holds raw device pointers into `keep` and `ws`: saved (alive + version-checked: an in-place change before backward
        # raises instead of silently giving wrong gradients) and re-read in backward before the pointers are used
        ctx.save_for_backward(ws, *[t for t in keep.values() if t is not None])
        ctx.desc, ctx.dims = d, (N, M)
        ctx.in_shapes = (image.shape, weights_sum.shape, depth.shape, weights.shape)
        ctx.in_dtypes = (image.dtype, weights_sum.dtype, depth.dtype, weights.dtype)
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), out_rgbs, out_normals, losses

    @staticmethod
    def backward(ctx, g_loss, g_rgbs, g_normals, _unused):
        N, M = ctx.dims
        ws = ctx.saved_tensors[0]
        dev = ws.device
        gl = _f32(g_loss, dev)              # stays on the device: the kernels read it (None = 1)
        g_rgbs, g_normals = _f32(g_rgbs, dev), _f32(g_normals, dev)
        g_image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        g_ws = torch.empty(N, dtype=torch.float32, device=dev)
        g_depth = torch.empty(N, dtype=torch.float32, device=dev)
        g_weights = torch.empty(M, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call('mve_recon_loss_backward', ctypes.byref(ctx.desc), _lib.ptr(ws), ws.numel(), _lib.ptr(g_rgbs), _lib.ptr(g_normals), _lib.ptr(gl),
                      _lib.ptr(g_image), _lib.ptr(g_ws), _lib.ptr(g_depth), _lib.ptr(g_weights), _lib.stream_ptr(dev))
        outs = [g.reshape(s).to(t) for g, s, t in zip((g_image, g_ws, g_depth, g_weights), ctx.in_shapes, ctx.in_dtypes)]
        return (*outs, None, None)


def nerf_optim_loss(image, weights_sum, depth, weights, ts, target_rgbs, target_m_blur, target_dir, patch_w, patch_lights, *, target_n=None,
                    target_depth=None, tonemapping=None, shaded=True, is_init=False, ambient_light=0.2, bg_color=1.0, normal_bg=(0.5, 0.5, 1.0),
                    pixel_loss_weight=1.2, normal_reg_weight=0.0, depth_weight=0.0, entropy_weight=0.0, bg_width=0.015):
    """image [N, 3] / weights_sum [N] / depth [N] / weights [M]: `outputs` of the decoder for N = P * ps * ps rays (patch-major), ts [M, 2];
    target_rgbs [P, ps, ps, 3], target_m_blur [P, ps, ps, 1], target_dir [P, ps, ps, 3], optional target_n [P, ps, ps, 3] and
    target_depth [P, ps, ps, 1]; patch_w [P], patch_lights [P, 3]; tonemapping: `mvedit_amd.tonemapping.Tonemapping` or None.
    -> dict(loss, pixel_rgb_loss, alphas_loss, normal_reg_loss, depth_loss, entropy_loss (the parts detached), out_rgbs [P, ps, ps, 3],
    out_normals [P, ps, ps, 3]); `loss`, `out_rgbs` and `out_normals` are differentiable w.r.t. image, weights_sum, depth, weights."""
    consts = dict(ts=ts, target_rgbs=target_rgbs, target_m=target_m_blur, target_dir=target_dir, target_n=target_n, target_depth=target_depth,
                  patch_w=patch_w, patch_lights=patch_lights, lut_x=None if tonemapping is None else tonemapping.lut_x,
                  lut_y=None if tonemapping is None else tonemapping.lut_y)
    hyper = dict(shaded=bool(shaded), is_init=bool(is_init), ambient_light=float(ambient_light), bg_color=float(bg_color),
                 normal_bg=tuple(float(v) for v in normal_bg), pixel_loss_weight=float(pixel_loss_weight), normal_reg_weight=float(normal_reg_weight),
                 depth_weight=float(depth_weight), entropy_weight=float(entropy_weight), bg_width=float(bg_width))
    loss, out_rgbs, out_normals, parts = _ReconLossFn.apply(image, weights_sum, depth, weights, consts, hyper)
    res = dict(zip(PARTS, parts.unbind(0)))
    res.update(loss=loss, out_rgbs=out_rgbs, out_normals=out_normals)
    return res


class _MeshDesc(ctypes.Structure):
    """MveMeshLossDesc (include/mvedit_amd.h)"""
    _fields_ = ([(n, ctypes.c_int32) for n in ('n', 'size', 'mesh_is_simplified')] + [('normal_bg', ctypes.c_float * 3)]
                + [(n, ctypes.c_float) for n in ('pixel_loss_weight', 'normal_reg_weight')]
                + [(n, ctypes.c_void_p) for n in ('d_rgba', 'd_normal', 'd_depth', 'd_target_dir', 'd_target_rgbs', 'd_target_m_erode', 'd_target_m_blur',
                                                  'd_target_n', 'd_view_w')])


class _MeshLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgba, normal, consts, hyper):
        dev = rgba.device
        assert rgba.is_cuda, 'native path: CUDA tensors only'
        n, S = consts['target_rgbs'].shape[:2]
        N = n * S * S
        keep = dict(rgba=_f32(rgba, dev).reshape(N, 4), normal=_f32(normal, dev).reshape(N, 3), **{k: _f32(v, dev) for k, v in consts.items()})
        assert keep['depth'].numel() == N and keep['target_dir'].numel() == 3 * N and keep['view_w'].numel() == n
        d = _MeshDesc(n=n, size=S, mesh_is_simplified=int(hyper['simplified']), normal_bg=(ctypes.c_float * 3)(*hyper['normal_bg']),
                      pixel_loss_weight=hyper['pixel_loss_weight'], normal_reg_weight=hyper['normal_reg_weight'])
        for field, key in (('d_rgba', 'rgba'), ('d_normal', 'normal'), ('d_depth', 'depth'), ('d_target_dir', 'target_dir'),
                           ('d_target_rgbs', 'target_rgbs'), ('d_target_m_erode', 'target_m_erode'), ('d_target_m_blur', 'target_m_blur'),
                           ('d_target_n', 'target_n'), ('d_view_w', 'view_w')):
            setattr(d, field, None if keep[key] is None else keep[key].data_ptr())
        ws = torch.empty(_lib.raw('mve_mesh_loss_workspace_bytes')(n, S), dtype=torch.uint8, device=dev)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        out_rgbs = torch.empty(n, S, S, 3, dtype=torch.float32, device=dev)
        out_normals = torch.empty(n, S, S, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call('mve_mesh_loss_forward', ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.ptr(losses), _lib.ptr(out_rgbs), _lib.ptr(out_normals),
                      _lib.stream_ptr(dev))
        ctx.save_for_backward(ws, *[t for t in keep.values() if t is not None])
        ctx.desc, ctx.N = d, N
        ctx.in_shapes, ctx.in_dtypes = (rgba.shape, normal.shape), (rgba.dtype, normal.dtype)
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), out_rgbs, out_normals, losses

    @staticmethod
    def backward(ctx, g_loss, g_rgbs, g_normals, _unused):
        ws = ctx.saved_tensors[0]
        dev = ws.device
        gl, g_rgbs, g_normals = _f32(g_loss, dev), _f32(g_rgbs, dev), _f32(g_normals, dev)
        g_rgba = torch.empty(ctx.N, 4, dtype=torch.float32, device=dev)
        g_normal = torch.empty(ctx.N, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call('mve_mesh_loss_backward', ctypes.byref(ctx.desc), _lib.ptr(ws), ws.numel(), _lib.ptr(g_rgbs), _lib.ptr(g_normals),
                      _lib.ptr(gl), _lib.ptr(g_rgba), _lib.ptr(g_normal), _lib.stream_ptr(dev))
        return g_rgba.reshape(ctx.in_shapes[0]).to(ctx.in_dtypes[0]), g_normal.reshape(ctx.in_shapes[1]).to(ctx.in_dtypes[1]), None, None


def mesh_optim_loss(rgba, normal, depth, target_rgbs, target_m_erode, target_m_blur, target_dir, view_w, *, target_n=None,
                    mesh_is_simplified=False, normal_bg=(0.5, 0.5, 1.0), pixel_loss_weight=1.2, normal_reg_weight=0.0):
    """Image-space part of one mesh optimisation iteration (lib/pipelines/mvedit_3d_pipeline.py:745-782): rgba [n, S, S, 4], normal
    [n, S, S, 3], depth [n, S, S] = `render_out['rgba' / 'normal' / 'depth'].squeeze(0)`; targets [n, S, S, C]; view_w [n] = cam_weights /
    cam_weights_mean.  -> dict(loss (= pixel_rgb_loss + alphas_loss + normal_reg_loss), the parts (detached), out_rgbs, out_normals);
    differentiable w.r.t. rgba and normal.  The mesh regularisers of the same sum are `mesh_ops.mesh_regularizers`."""
    consts = dict(depth=depth, target_rgbs=target_rgbs, target_m_erode=target_m_erode, target_m_blur=target_m_blur, target_dir=target_dir,
                  target_n=target_n, view_w=view_w)
    hyper = dict(simplified=bool(mesh_is_simplified), normal_bg=tuple(float(v) for v in normal_bg), pixel_loss_weight=float(pixel_loss_weight),
                 normal_reg_weight=float(normal_reg_weight))
    loss, out_rgbs, out_normals, parts = _MeshLossFn.apply(rgba, normal, consts, hyper)
    res = dict(zip(('loss', 'pixel_rgb_loss', 'alphas_loss', 'normal_reg_loss'), parts.unbind(0)))
    res.update(loss=loss, out_rgbs=out_rgbs, out_normals=out_normals)
    return res
