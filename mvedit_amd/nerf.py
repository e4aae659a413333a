"""Host-side mirror of the reference's NeRF render seam on top of csrc/nerf.hip.

    nerf.render(decoder, code, density_bitfield, h, w, intrinsics[1,b,4], poses[1,b,3,4],
                cfg=dict(return_rgba=True, compute_normal=True, dt_gamma_scale=...), perturb, normal_bg)
        -> (rgba [1,b,h,w,4], depth [1,b,h,w], normal [1,b,h,w,3], normal_fg)          lib/models/autoencoders/base_nerf.py:489-556

`INGPDecoderParams` holds what `iNGPDecoder` (lib/models/decoders/ingp_decoder.py:44-120) owns: the tinycudann hash table
and the two Linear layers.  Everything is float32, as in the reference (raymarching `custom_fwd(cast_inputs=float32)`,
tcnn dtype forced to float32 at ingp_decoder.py:73).  No torch fallback exists for any of the calls.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib


def grid_meta(n_levels=12, base_resolution=16, max_resolution=320, bound=1.0, log2_hashmap_size=19):
    """tiny-cuda-nn HashGrid level table: (scale f32, resolution, row offset, rows) per level + total rows."""
    pls = np.exp2(np.log2(max_resolution * bound / base_resolution) / (n_levels - 1))     # ingp_decoder.py:71
    log2_pls = np.float32(np.log2(np.float32(pls)))
    meta, off = [], 0
    for lvl in range(n_levels):
        scale = np.float32(np.exp2(np.float32(lvl) * log2_pls, dtype=np.float32) * np.float32(base_resolution) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        n = min((res ** 3 + 7) // 8 * 8, 1 << log2_hashmap_size)
        meta.append((float(scale), res, off, n))
        off += n
    return meta, off


class _PointDecodeFn(torch.autograd.Function):
    """point_decode as a differentiable op w.r.t. the decoder parameters: native forward, native backward (the forward is
    recomputed inside the backward kernel, so nothing but the inputs is saved).  Gradients w.r.t. the sample positions are not
    provided (the reference needs them only for compute_normal in the training branch)."""

    @staticmethod
    def forward(ctx, dec, xyzs, table, w1, b1, w2, b2):
        ctx.dec = dec
        ctx.save_for_backward(xyzs)
        sigmas, rgbs = dec.point_decode(xyzs)
        return sigmas, rgbs

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        (xyzs,) = ctx.saved_tensors
        g = ctx.dec.point_decode_backward(xyzs, g_sigma.contiguous(), g_rgb.contiguous())
        return None, None, g['table'], g['w1'], g['b1'], g['w2'], g['b2']


class INGPDecoderParams:
    """Device-resident state of an iNGPDecoder (hash table + MLP) in the layout the kernels read."""

    def __init__(self, table, w1, b1, w2, b2, n_levels=12, max_resolution=320, bound=1.0, blob_density=1.0, blob_radius=0.2,
                 sigmoid_saturation=0.001, min_near=0.2, max_steps=1024, device='cuda'):
        f = lambda t: torch.as_tensor(t, dtype=torch.float32).to(device).contiguous()
        self.table, self.w1, self.b1, self.w2, self.b2 = f(table), f(w1), f(b1), f(w2), f(b2)
        self.n_levels, self.max_resolution, self.bound = n_levels, max_resolution, float(bound)
        self.blob_density, self.blob_radius, self.sigmoid_saturation = blob_density, blob_radius, sigmoid_saturation
        self.min_near, self.max_steps = min_near, max_steps
        self.hidden = self.w1.shape[0]
        meta, rows = grid_meta(n_levels, 16, max_resolution, bound)
        assert self.table.shape == (rows, 2), (self.table.shape, rows)
        assert self.w1.shape == (self.hidden, 2 * n_levels) and self.w2.shape == (4, self.hidden)
        self._scale = (ctypes.c_float * n_levels)(*[m[0] for m in meta])
        self._res = (ctypes.c_uint32 * n_levels)(*[m[1] for m in meta])
        self._off = (ctypes.c_uint32 * n_levels)(*[m[2] for m in meta])
        self._size = (ctypes.c_uint32 * n_levels)(*[m[3] for m in meta])
        self.aabb = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32, device=device)

    @property
    def device(self):
        return self.table.device

    def _grid_args(self):
        return (_lib.ptr(self.table), self.n_levels, self._scale, self._res, self._off, self._size)

    def _mlp_args(self):
        return (_lib.ptr(self.w1), _lib.ptr(self.b1), _lib.ptr(self.w2), _lib.ptr(self.b2), self.hidden)

    # iNGPDecoder.point_decode / point_density_decode ---------------------------------------------------------
    def point_decode(self, xyzs, density_only=False):
        xyzs = xyzs.to(self.device, torch.float32).contiguous().view(-1, 3)
        M = xyzs.shape[0]
        sigmas = torch.empty(M, dtype=torch.float32, device=self.device)
        rgbs = None if density_only else torch.empty(M, 3, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.call('mve_hashgrid_mlp_decode', _lib.ptr(xyzs), M, *self._grid_args(), *self._mlp_args(), self.bound,
                      self.blob_density, self.blob_radius, self.sigmoid_saturation, _lib.ptr(sigmas), _lib.ptr(rgbs),
                      _lib.stream_ptr(self.device))
        return sigmas, rgbs

    # reconstruct step: gradients and optimiser -------------------------------------------------
    def parameters(self):
        return dict(table=self.table, w1=self.w1, b1=self.b1, w2=self.w2, b2=self.b2)

    def point_decode_backward(self, xyzs, grad_sigmas, grad_rgbs=None, grads=None):
        """Gradients of sum(grad_sigmas * sigma) + sum(grad_rgbs * rgb) w.r.t. parameters().  `grads` (a dict from a previous
        call) is re-used: its table gradient is accumulated into, the MLP gradients are overwritten."""
        xyzs = xyzs.to(self.device, torch.float32).contiguous().view(-1, 3)
        M = xyzs.shape[0]
        gs = grad_sigmas.to(self.device, torch.float32).contiguous().view(-1)
        gr = grad_rgbs.to(self.device, torch.float32).contiguous().view(-1, 3) if grad_rgbs is not None else None
        if grads is None:
            grads = {k: torch.zeros_like(v) for k, v in self.parameters().items()}
        nbytes = _lib.raw('mve_hashgrid_mlp_backward_workspace_bytes')(M, self.n_levels)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.call('mve_hashgrid_mlp_backward', _lib.ptr(xyzs), M, *self._grid_args(), *self._mlp_args(), self.bound, self.blob_density,
                      self.blob_radius, self.sigmoid_saturation, _lib.ptr(gs), _lib.ptr(gr), _lib.ptr(grads['table']), _lib.ptr(grads['w1']),
                      _lib.ptr(grads['b1']), _lib.ptr(grads['w2']), _lib.ptr(grads['b2']), _lib.ptr(ws), nbytes, _lib.stream_ptr(self.device))
        return grads

    def point_decode_autograd(self, xyzs):
        """point_decode whose outputs carry autograd history w.r.t. parameters(): mark the parameter tensors with
        requires_grad_(True) and use any torch optimiser on them, as the reference's nerf_optim does with tinycudann modules."""
        return _PointDecodeFn.apply(self, xyzs, self.table, self.w1, self.b1, self.w2, self.b2)

    def adam_step(self, grads, state, lr=1e-2, betas=(0.9, 0.999), eps=1e-15):
        """One torch.optim.Adam step on every parameter, in place.  `state` is a dict the caller keeps between steps."""
        state['step'] = state.get('step', 0) + 1
        with torch.cuda.device(self.device):
            for k, w in self.parameters().items():
                if k not in state:
                    state[k] = (torch.zeros_like(w), torch.zeros_like(w))
                m1, m2 = state[k]
                _lib.call('mve_adam_step', _lib.ptr(w), _lib.ptr(grads[k]), _lib.ptr(m1), _lib.ptr(m2), w.numel(), float(lr), float(betas[0]),
                          float(betas[1]), float(eps), int(state['step']), _lib.stream_ptr(self.device))
        return state

    # VolumeRenderer.forward, eval branch ------------------------------------------------------------------------
    def render_rays(self, rays_o, rays_d, density_bitfield, grid_size, dt_gamma=0.0, T_thresh=1e-2, return_counts=False):
        rays_o = rays_o.to(self.device, torch.float32).contiguous().view(-1, 3)
        rays_d = rays_d.to(self.device, torch.float32).contiguous().view(-1, 3)
        N = rays_o.shape[0]
        ws = torch.empty(N, dtype=torch.float32, device=self.device)
        depth = torch.empty(N, dtype=torch.float32, device=self.device)
        image = torch.empty(N, 3, dtype=torch.float32, device=self.device)
        counts = torch.empty(N, dtype=torch.int32, device=self.device) if return_counts else None
        bits = density_bitfield.to(self.device).contiguous()
        with torch.cuda.device(self.device):
            _lib.call('mve_nerf_render_rays', _lib.ptr(rays_o), _lib.ptr(rays_d), N, _lib.ptr(bits), int(grid_size),
                      _lib.ptr(self.aabb), self.bound, self.min_near, float(dt_gamma), int(self.max_steps), float(T_thresh),
                      *self._grid_args(), *self._mlp_args(), self.blob_density, self.blob_radius, self.sigmoid_saturation,
                      _lib.ptr(ws), _lib.ptr(depth), _lib.ptr(image), _lib.ptr(counts), _lib.stream_ptr(self.device))
        return (ws, depth, image, counts) if return_counts else (ws, depth, image)


def camera_rays(intrinsics, poses, h, w):
    """intrinsics [b,4], poses [b,3,4] -> rays_o, rays_d [b*h*w,3], dir_norm [b,h,w]  (geometry_utils.py:18-55)."""
    intrinsics = intrinsics.float().contiguous()
    poses = poses[..., :3, :4].float().contiguous()
    b = intrinsics.shape[0]
    dev = intrinsics.device
    rays_o = torch.empty(b * h * w, 3, dtype=torch.float32, device=dev)
    rays_d = torch.empty_like(rays_o)
    dir_norm = torch.empty(b, h, w, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call('mve_camera_rays', _lib.ptr(intrinsics), _lib.ptr(poses), b, h, w, _lib.ptr(rays_o), _lib.ptr(rays_d),
                  _lib.ptr(dir_norm), _lib.stream_ptr(dev))
    return rays_o, rays_d, dir_norm


def depth_to_normal(depth, intrinsics, alpha=None, normal_bg=(0.5, 0.5, 1.0)):
    """depth [b,h,w] (1/z); alpha [b,h,w] view (any element stride) or None -> (normal_fg, normal) [b,h,w,3]."""
    depth = depth.float().contiguous()
    b, h, w = depth.shape
    dev = depth.device
    intrinsics = intrinsics.float().contiguous()
    normal_fg = torch.empty(b, h, w, 3, dtype=torch.float32, device=dev)
    normal = torch.empty_like(normal_fg) if alpha is not None else None
    stride = 1
    if alpha is not None:
        assert alpha.shape == depth.shape and alpha.dtype == torch.float32
        stride = alpha.stride(-1)
        assert alpha.stride(-2) == w * stride and alpha.stride(-3) == h * w * stride
    bg = (ctypes.c_float * 3)(*normal_bg)
    with torch.cuda.device(dev):
        _lib.call('mve_depth_to_normal', _lib.ptr(depth), _lib.ptr(alpha), stride, _lib.ptr(intrinsics), b, h, w, bg,
                  _lib.ptr(normal_fg), _lib.ptr(normal), _lib.stream_ptr(dev))
    return normal_fg, normal


def normalize_depth(depths, alphas, far_depth=0.25, alpha_clip=0.5, eps=1e-5):
    """geometry_utils.py:151-168.  depths [N,H,W], alphas [N,H,W,1]."""
    depths = depths.float().contiguous()
    a = alphas.float().reshape(depths.shape).contiguous()
    out = torch.empty_like(depths)
    n, hh, ww = depths.shape
    with torch.cuda.device(depths.device):
        _lib.call('mve_normalize_depth', _lib.ptr(depths), _lib.ptr(a), n, hh * ww, far_depth, alpha_clip, eps, _lib.ptr(out),
                  _lib.stream_ptr(depths.device))
    return out


class NeRFRenderer:
    """Stands in for `BaseNeRF` at the `nerf.render(...)` seam (lib/pipelines/mvedit_3d_pipeline.py:1363-1371)."""

    def __init__(self, grid_size=128, bg_color=1.0):
        self.grid_size = grid_size
        self.bg_color = bg_color

    def render(self, decoder, code, density_bitfield, h, w, intrinsics, poses, cfg=dict(), bg_color=None, perturb=False,
               normal_bg=(0.5, 0.5, 1.0)):
        assert not perturb, 'eval rendering is deterministic (perturb=False) in the reference pipelines'
        assert intrinsics.dim() == 3 and intrinsics.shape[0] == 1, 'one scene (num_scenes = 1), as in the MVEdit pipelines'
        bg_color = self.bg_color if bg_color is None else bg_color
        intr, pose = intrinsics[0].float(), poses[0].float()
        b = intr.shape[0]
        dt_gamma = float(cfg.get('dt_gamma_scale', 0.0) * 2 / (intr[:, 0] + intr[:, 1]).mean())      # base_nerf.py:503-504
        rays_o, rays_d, dir_norm = camera_rays(intr, pose, h, w)
        bits = density_bitfield.reshape(-1)
        ws, depth, image = decoder.render_rays(rays_o, rays_d, bits, self.grid_size, dt_gamma)
        depth = depth.view(b, h, w)
        if cfg.get('inverse_z_depth', True):
            depth = depth * dir_norm                                                                   # 1/r -> 1/z
        if cfg.get('return_rgba', False):
            out_image = torch.cat([image, ws[:, None]], dim=-1).view(1, b, h, w, 4)
        else:
            out_image = (image + bg_color * (1 - ws[:, None])).view(1, b, h, w, 3)
        if cfg.get('compute_normal', False):
            assert cfg.get('inverse_z_depth', True) and cfg.get('return_rgba', False)
            normal_fg, normal = depth_to_normal(depth, intr, out_image[0, ..., 3], normal_bg)
            return out_image, depth[None], normal[None], normal_fg[None]
        return out_image, depth[None]


class VolumeRenderer:
    """Mirror of the reference's VolumeRenderer for ONE scene with an iNGP decoder: the eval branch (fused, see
    INGPDecoderParams.render_rays), the train-branch FORWARD (march -> density -> cull -> decode -> composite,
    lib/models/decoders/base_volume_renderer.py:207-262) and `update_extra_state` (:105-177).
    With requires_grad decoder parameters the training forward is differentiable (native decode / composite backward)."""

    def __init__(self, decoder, weight_culling_th=1e-3):
        self.decoder = decoder
        self.bound, self.min_near, self.max_steps = decoder.bound, decoder.min_near, decoder.max_steps
        self.weight_culling_th = weight_culling_th
        self.training = False

    def update_extra_state(self, density_grid, density_bitfield, iter_density, density_thresh=0.01, decay=0.9, S=128):
        """density_grid [1, H^3] f32 and density_bitfield [1, H^3/8] u8 are updated IN PLACE; returns iter_density + 1 and the
        threshold used.  Random draws (jitter, partial-update cells) come from torch's generator, as in the reference."""
        from . import raymarching as rm
        dec, dev = self.decoder, self.decoder.device
        assert density_grid.shape[0] == 1 and density_grid.dtype == torch.float32 and density_grid.is_contiguous()
        n_cells = density_grid.shape[-1]
        H = int(round(n_cells ** (1.0 / 3.0)))
        tmp = torch.full_like(density_grid, -1)
        if iter_density < 16:                                         # full update, one launch over the whole grid
            N, coords = n_cells, None
        else:                                                         # partial update: N random cells + N occupied cells
            N = n_cells // 4
            coords = torch.randint(0, H, (N, 3), device=dev)
            occ = torch.nonzero(density_grid[0] > 0).squeeze(-1)
            occ = occ[torch.randint(0, occ.shape[0], [N], dtype=torch.long, device=dev)]
            coords = torch.cat([coords.int(), rm.morton3D_invert(occ.int())], dim=0).contiguous()
            N = coords.shape[0]
        noise = torch.rand(N, 3, dtype=torch.float32, device=dev)
        xyzs = torch.empty(N, 3, dtype=torch.float32, device=dev)
        indices = torch.empty(N, dtype=torch.int32, device=dev)
        mean = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = torch.empty(_lib.raw('mve_density_grid_scratch_bytes')(n_cells), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call('mve_density_grid_points', _lib.ptr(coords), _lib.ptr(noise), N, H, self.bound, _lib.ptr(xyzs),
                      _lib.ptr(indices), _lib.stream_ptr(dev))
            sigmas, _ = dec.point_decode(xyzs, density_only=True)
            _lib.call('mve_density_grid_update', _lib.ptr(density_grid), _lib.ptr(tmp), n_cells, _lib.ptr(sigmas), _lib.ptr(indices),
                      N, float(decay), _lib.ptr(mean), _lib.ptr(scratch), _lib.stream_ptr(dev))
        thresh = min(float(mean.item()), density_thresh)              # the reference reads the mean on the host too (:171-174)
        rm.packbits(density_grid, thresh, density_bitfield)
        return iter_density + 1, thresh

    def forward(self, rays_o, rays_d, density_bitfield, grid_size, dt_gamma=0.0, perturb=False, noises=None):
        """rays_o, rays_d [N,3] of one scene.  Returns the reference's result dict (weights, weights_sum, depth, image, rays,
        ts; lists of one entry where the reference returns per-scene lists)."""
        from . import raymarching as rm
        dec = self.decoder
        if not self.training:
            ws, depth, image = dec.render_rays(rays_o, rays_d, density_bitfield, grid_size, dt_gamma)
            return dict(weights=None, weights_sum=[ws], depth=[depth], image=[image], rays=None, normal=[None], ts=None)
        nears, fars = rm.near_far_from_aabb(rays_o, rays_d, dec.aabb, self.min_near)
        xyzs, dirs, ts, rays = rm.march_rays_train(rays_o, rays_d, self.bound, density_bitfield, 1, grid_size, nears, fars,
                                                   perturb=perturb, dt_gamma=float(dt_gamma), max_steps=self.max_steps, noises=noises)
        if self.weight_culling_th > 0:
            with torch.no_grad():                                      # base_volume_renderer.py:223
                sigmas, _ = dec.point_decode(xyzs, density_only=True)
                weights, _, _, _ = rm.batch_composite_rays_train(sigmas, sigmas.new_zeros(sigmas.shape[0], 3), [ts], [rays], [ts.shape[0]])
                xyzs, dirs, ts, rays = rm.cull_samples(weights, self.weight_culling_th, xyzs, dirs, ts, rays)
        # with requires_grad parameters the outputs carry autograd history (native backward for decode and composite)
        differentiable = torch.is_grad_enabled() and any(t.requires_grad for t in dec.parameters().values())
        sigmas, rgbs = dec.point_decode_autograd(xyzs) if differentiable else dec.point_decode(xyzs)
        weights, weights_sum, depth, image = rm.batch_composite_rays_train(sigmas, rgbs, [ts], [rays], [ts.shape[0]])
        return dict(weights=weights, weights_sum=weights_sum, depth=depth, image=image, rays=[rays], normal=None, ts=[ts])
