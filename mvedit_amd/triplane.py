"""Mirrors of the reference's tri-plane radiance decoders for one scene, forward only: `TriPlaneDecoder.point_decode` /
`point_density_decode` (lib/models/decoders/triplane_decoder.py:107-204) and `TriPlaneiNGPDecoder.point_decode`
(lib/models/decoders/triplane_ingp_decoder.py:142-212) -> mve_triplane_decode (csrc/triplane.hip): plane fetches, the base / density /
colour MLPs, the SH direction encoding and (iNGP variant) the hash-grid branch in ONE launch, one wave per 64 points.
Supported topology = the classes' defaults (one Linear in base_net / density_net / ingp_base_net, Linear-act-Linear colour net over
cat[act(base), SH_4(dir)], dir_layers=None, no scene_base / code dropout); anything else raises.
Gradients: `point_decode_autograd` carries autograd history w.r.t. the code planes and `parameters()` (every Linear and the hash table)
through mve_triplane_backward -- what nerf_optim (lib/pipelines/mvedit_3d_pipeline.py:507-633) needs to optimise a TriPlaneiNGPDecoder
scene.  Plain `point_decode` is forward only and refuses tensors that require grad instead of silently cutting the graph.  No gradient
w.r.t. xyzs / dirs (the sample positions are not optimised)."""
import ctypes

import numpy as np
import torch

from . import _lib
from .nerf import grid_meta

_ACT = dict(relu=0, silu=1, softplus=2, trunc_exp=3)


class _Desc(ctypes.Structure):
    _fields_ = [('d_xyz', ctypes.c_void_p), ('d_dirs', ctypes.c_void_p), ('d_code', ctypes.c_void_p),
                ('N', ctypes.c_int32), ('C', ctypes.c_int32), ('h', ctypes.c_int32), ('w', ctypes.c_int32),
                ('axes', ctypes.c_int32 * 6), ('flip_z', ctypes.c_int32),
                ('d_base_wT', ctypes.c_void_p), ('d_base_b', ctypes.c_void_p), ('hidden', ctypes.c_int32), ('hidden2', ctypes.c_int32),
                ('d_ingp_wT', ctypes.c_void_p), ('d_ingp_b', ctypes.c_void_p), ('d_table', ctypes.c_void_p),
                ('n_levels', ctypes.c_int32), ('bound', ctypes.c_float),
                ('level_scale', ctypes.c_void_p), ('level_res', ctypes.c_void_p), ('level_offset', ctypes.c_void_p), ('level_size', ctypes.c_void_p),
                ('d_dens_w', ctypes.c_void_p), ('d_dens_b', ctypes.c_void_p), ('d_col1_wT', ctypes.c_void_p), ('d_col1_b', ctypes.c_void_p),
                ('d_col2_w', ctypes.c_void_p), ('d_col2_b', ctypes.c_void_p),
                ('activation', ctypes.c_int32), ('sigma_activation', ctypes.c_int32), ('sigmoid_saturation', ctypes.c_float),
                ('d_sigmas', ctypes.c_void_p), ('d_rgbs', ctypes.c_void_p)]


class _Grads(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('d_code', 'd_table', 'd_base_w', 'd_base_b', 'd_ingp_w', 'd_ingp_b', 'd_dens_w', 'd_dens_b',
                                                'd_col1_w', 'd_col1_b', 'd_col2_w', 'd_col2_b')]


class TriPlaneDecoder:
    """plane_cfg / flip_z / activation / sigma_activation / sigmoid_saturation as the reference constructor; weights come in through
    load_state_dict with the reference module's names (base_net.0.*, density_net.0.*, color_net.0.*, color_net.2.*)."""

    def __init__(self, plane_cfg=('xy', 'xz', 'yz'), activation='silu', sigma_activation='trunc_exp', sigmoid_saturation=0.001, flip_z=False,
                 bound=1.0, device='cuda'):
        assert len(plane_cfg) == 3 and all(len(p) == 2 and set(p) <= set('xyz') for p in plane_cfg), plane_cfg
        self.axes = [dict(x=0, y=1, z=2)[a] for p in plane_cfg for a in p]
        self.activation, self.sigma_activation = _ACT[activation.lower()], _ACT[sigma_activation.lower()]
        assert self.activation < 3
        self.sigmoid_saturation, self.flip_z, self.bound, self.device = float(sigmoid_saturation), bool(flip_z), float(bound), torch.device(device)
        self.w = {}
        self.hash = None

    # ---- parameters ------------------------------------------------------------------------------------------------------------
    _NAMES = {'base_net.0.weight': ('base_wT', True), 'base_net.0.bias': ('base_b', False),
              'density_net.0.weight': ('dens_w', False), 'density_net.0.bias': ('dens_b', False),
              'color_net.0.weight': ('col1_wT', True), 'color_net.0.bias': ('col1_b', False),
              'color_net.2.weight': ('col2_w', False), 'color_net.2.bias': ('col2_b', False),
              'ingp_base_net.0.weight': ('ingp_wT', True), 'ingp_base_net.0.bias': ('ingp_b', False)}

    def load_state_dict(self, sd):
        self.params = {}                 # reference name -> fp32 tensor in the reference's own layout (the leaves of point_decode_autograd)
        for name, t in sd.items():
            if name in self._NAMES:
                self.params[name] = t.detach().to(self.device, torch.float32).clone().contiguous()
            elif name == 'encoder.params':
                self.params[name] = t.detach().to(self.device, torch.float32).clone().contiguous()
        self._pack()
        for name in sd:
            if name in self._NAMES or name == 'encoder.params':
                continue
            if name.startswith(('base_net.', 'density_net.', 'color_net.', 'dir_net.', 'ingp_base_net.', 'scene_base')):
                raise NotImplementedError(f'{name}: only the default tri-plane decoder topology is built (one Linear per base / density / ingp '
                                          'net, Linear-act-Linear colour net, no dir_net / scene_base)')
        missing = [k for k in ('base_wT', 'base_b', 'dens_w', 'dens_b', 'col1_wT', 'col1_b', 'col2_w', 'col2_b') if k not in self.w]
        if missing:
            raise KeyError(f'tri-plane decoder: missing parameters {missing}')
        H, H2 = self.w['base_wT'].shape[1], self.w['col1_wT'].shape[1]
        assert self.w['col1_wT'].shape[0] == H + 16 and self.w['dens_w'].numel() == H and self.w['col2_w'].shape == (3, H2), 'unexpected layer widths'
        return self

    def _pack(self, tensors=None):
        """kernel-side copies of the parameters: weight matrices transposed to [in][out] (one input's fan-out = one contiguous scalar load).
        Without `tensors` the copies are rebuilt only when a parameter changed -- not on every call of the NeRF sampling hot path.  "Changed" is
        read off (object identity, storage address, torch's version counter): optimiser steps, in-place ops and load_state_dict bump the counter or
        replace the tensor.  Edits THROUGH `.data` (`p.data.copy_()`, `p.data.clamp_()`) and `set_` bump nothing: call `invalidate()` after them
        (ADVICE round 4).  Every packed tensor is a COPY (`.clone()`), never an alias of the parameter, so a missed edit leaves all of them equally
        stale instead of a mix of fresh aliases and stale transposes."""
        src = self.params if tensors is None else tensors
        if tensors is None:
            sig = tuple((n, id(t), t.data_ptr(), t._version) for n, t in src.items())
            if getattr(self, '_packed_sig', None) == sig:
                return
            self._packed_sig = sig
        else:
            self._packed_sig = None
        for name, t in src.items():
            if name in self._NAMES:
                key, transpose = self._NAMES[name]
                t = t.detach()
                self.w[key] = t.t().clone(memory_format=torch.contiguous_format) if transpose else t.clone()      # (.t().contiguous() aliases a [1, N] weight)
            elif name == 'encoder.params':
                self.w['table'] = t.detach().reshape(-1, 2).clone()

    def invalidate(self):
        """Forget the packed copies: the next call repacks from parameters().  Needed only after edits torch's version counter does not see
        (`.data` writes, `set_`)."""
        self._packed_sig = None

    def parameters(self):
        """reference name -> tensor (fp32, the reference module's layout); mark them requires_grad_(True) and optimise them with any torch optimiser"""
        return self.params

    # ---- forward ---------------------------------------------------------------------------------------------------------------
    def point_decode_autograd(self, xyzs, dirs, code, density_only=False):
        """point_decode whose outputs carry autograd history w.r.t. `code` and parameters() (native backward, mve_triplane_backward)"""
        assert len(xyzs) == 1 and code.shape[0] == 1, 'one scene per call (as every MVEdit pipeline)'
        names = sorted(self.params)
        d = None if (density_only or dirs is None) else dirs[0]
        sig, rgb = _TriDecodeFn.apply(self, names, xyzs[0], d, code, *[self.params[n] for n in names])
        return sig, (None if d is None else rgb), [sig.shape[0]]

    def point_decode(self, xyzs, dirs, code, density_only=False, use_2nd_order=False):
        """xyzs: [1, (N, 3)] (list or tensor), dirs likewise or None, code [1, 3, C, h, w] -> (sigmas [N], rgbs [N,3] | None, [N])"""
        assert not use_2nd_order, 'lib/ops/cuda_gridsample (second-order grid_sample) is not part of this engine'
        assert len(xyzs) == 1 and code.shape[0] == 1, 'one scene per call (as every MVEdit pipeline)'
        xyz = xyzs[0]
        d = None if (density_only or dirs is None) else dirs[0]
        # forward only: PARAMETERS that are marked trainable (nerf_optim's optimiser holds them) are read as constants; an input that requires
        # grad is refused -- its gradient would silently be missing
        for t in (xyz, d, code):
            if t is not None and torch.is_grad_enabled() and t.requires_grad:
                raise NotImplementedError('tri-plane decoders: point_decode is the forward only -- use point_decode_autograd for gradients '
                                          '(or call under torch.no_grad())')
        self._pack()
        return self._forward(xyz, d, code)

    def _prep(self, xyz, d, code):
        xyz = xyz.detach().to(self.device, torch.float32).reshape(-1, 3).contiguous()
        N = xyz.shape[0]
        if d is not None:
            d = d.detach().to(self.device, torch.float32).reshape(-1, 3).contiguous()
            assert d.shape[0] == N
        _, _, C, h, w = code.shape
        cl = code[0].detach().to(self.device, torch.float32).permute(0, 2, 3, 1).contiguous()      # [3][h][w][C]
        W = self.w
        assert W['base_wT'].shape[0] == 3 * C, (W['base_wT'].shape, C)
        return xyz, d, cl, N, C, h, w

    def _desc(self, xyz, d, cl, N, C, h, w, sig, rgb):
        W = self.w
        ds = _Desc()
        ds.d_xyz, ds.d_dirs, ds.d_code = xyz.data_ptr(), (d.data_ptr() if d is not None else None), cl.data_ptr()
        ds.N, ds.C, ds.h, ds.w = N, C, h, w
        for k in range(6):
            ds.axes[k] = self.axes[k]
        ds.flip_z = int(self.flip_z)
        ds.d_base_wT, ds.d_base_b = W['base_wT'].data_ptr(), W['base_b'].data_ptr()
        ds.hidden, ds.hidden2 = W['base_wT'].shape[1], W['col1_wT'].shape[1]
        keep = []
        if self.hash is not None:
            meta = self.hash['meta']
            arr = (np.array([m[0] for m in meta], np.float32), np.array([m[1] for m in meta], np.uint32),
                   np.array([m[2] for m in meta], np.uint32), np.array([m[3] for m in meta], np.uint32))
            keep.append(arr)
            ds.d_ingp_wT, ds.d_ingp_b, ds.d_table = W['ingp_wT'].data_ptr(), W['ingp_b'].data_ptr(), W['table'].data_ptr()
            ds.n_levels, ds.bound = len(meta), self.bound
            ds.level_scale, ds.level_res, ds.level_offset, ds.level_size = (a.ctypes.data for a in arr)
        ds.d_dens_w, ds.d_dens_b = W['dens_w'].data_ptr(), W['dens_b'].data_ptr()
        ds.d_col1_wT, ds.d_col1_b, ds.d_col2_w, ds.d_col2_b = (W[k].data_ptr() for k in ('col1_wT', 'col1_b', 'col2_w', 'col2_b'))
        ds.activation, ds.sigma_activation, ds.sigmoid_saturation = self.activation, self.sigma_activation, self.sigmoid_saturation
        ds.d_sigmas, ds.d_rgbs = (sig.data_ptr() if sig is not None else None), (rgb.data_ptr() if rgb is not None else None)
        return ds, keep

    def _forward(self, xyz, d, code):
        xyz, d, cl, N, C, h, w = self._prep(xyz, d, code)
        sig = torch.empty(N, dtype=torch.float32, device=self.device)
        rgb = torch.empty(N, 3, dtype=torch.float32, device=self.device) if d is not None else None
        ds, keep = self._desc(xyz, d, cl, N, C, h, w, sig, rgb)
        with torch.cuda.device(self.device):
            _lib.call('mve_triplane_decode', ctypes.byref(ds), _lib.stream_ptr(self.device))
        del keep
        return sig, rgb, [N]

    def _backward(self, xyz, d, code, g_sig, g_rgb, need_code):
        """-> (grad code [1,3,C,h,w] or None, {reference name: gradient})"""
        xyz, d, cl, N, C, h, w = self._prep(xyz, d, code)
        dev, W = self.device, self.w
        H, H2 = W['base_wT'].shape[1], W['col1_wT'].shape[1]
        nl = len(self.hash['meta']) if self.hash is not None else 0
        z = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device=dev)
        g = {'base_net.0.weight': z(H, 3 * C), 'base_net.0.bias': z(H), 'density_net.0.weight': z(1, H), 'density_net.0.bias': z(1),
             'color_net.0.weight': z(H2, H + 16), 'color_net.0.bias': z(H2), 'color_net.2.weight': z(3, H2), 'color_net.2.bias': z(3)}
        if nl:
            g.update({'ingp_base_net.0.weight': z(H, 2 * nl), 'ingp_base_net.0.bias': z(H), 'encoder.params': torch.zeros_like(self.params['encoder.params'])})
        g_code = z(3, h, w, C) if need_code else None
        gs = _Grads()
        gs.d_code, gs.d_table = (g_code.data_ptr() if need_code else None), (g['encoder.params'].data_ptr() if nl else None)
        gs.d_base_w, gs.d_base_b = g['base_net.0.weight'].data_ptr(), g['base_net.0.bias'].data_ptr()
        if nl:
            gs.d_ingp_w, gs.d_ingp_b = g['ingp_base_net.0.weight'].data_ptr(), g['ingp_base_net.0.bias'].data_ptr()
        gs.d_dens_w, gs.d_dens_b = g['density_net.0.weight'].data_ptr(), g['density_net.0.bias'].data_ptr()
        gs.d_col1_w, gs.d_col1_b = g['color_net.0.weight'].data_ptr(), g['color_net.0.bias'].data_ptr()
        gs.d_col2_w, gs.d_col2_b = g['color_net.2.weight'].data_ptr(), g['color_net.2.bias'].data_ptr()
        ds, keep = self._desc(xyz, d, cl, N, C, h, w, None, None)
        nbytes = _lib.raw('mve_triplane_backward_workspace_bytes')(N, C, H, H2, nl)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        g_sig = g_sig.to(dev, torch.float32).contiguous()
        g_rgb = g_rgb.to(dev, torch.float32).contiguous() if (g_rgb is not None and d is not None) else None
        with torch.cuda.device(dev):
            _lib.call('mve_triplane_backward', ctypes.byref(ds), _lib.ptr(g_sig), _lib.ptr(g_rgb), ctypes.byref(gs), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
        del keep
        return (g_code.permute(0, 3, 1, 2)[None].contiguous() if need_code else None), g

    def point_density_decode(self, xyzs, code, **kwargs):
        sigmas, _, num_points = self.point_decode(xyzs, None, code, density_only=True, **kwargs)
        return sigmas, num_points


class TriPlaneiNGPDecoder(TriPlaneDecoder):
    """+ the hash-grid branch: base_x = base_net(plane features) + ingp_base_net(HashGrid((xyz + bound) / (2 bound))); the level table follows
    the constructor (triplane_ingp_decoder.py:102-114: 2 features per level, log2_hashmap_size 19, Smoothstep)."""

    def __init__(self, *args, base_resolution=16, max_resolution=320, n_levels=12, log2_hashmap_size=19, **kwargs):
        super().__init__(*args, **kwargs)
        meta, rows = grid_meta(n_levels, base_resolution, max_resolution, self.bound, log2_hashmap_size)
        self.hash = dict(meta=meta, rows=rows)

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        for k in ('ingp_wT', 'ingp_b', 'table'):
            if k not in self.w:
                raise KeyError(f'TriPlaneiNGPDecoder: missing parameter for {k}')
        assert self.w['table'].shape[0] == self.hash['rows'], (self.w['table'].shape, self.hash['rows'])
        assert self.w['ingp_wT'].shape == (2 * len(self.hash['meta']), self.w['base_wT'].shape[1])
        return self


class _TriDecodeFn(torch.autograd.Function):
    """(sigmas, rgbs) = decoder.point_decode with the native backward w.r.t. the code planes and the decoder's parameters"""

    @staticmethod
    def forward(ctx, dec, names, xyz, dirs, code, *params):
        dec._pack(dict(zip(names, params)))
        sig, rgb, _ = dec._forward(xyz, dirs, code)
        ctx.dec, ctx.names, ctx.has_dirs = dec, names, dirs is not None
        ctx.save_for_backward(xyz, dirs if dirs is not None else xyz.new_zeros(0), code, *params)
        if rgb is None:
            rgb = sig.new_zeros(0, 3)
            ctx.mark_non_differentiable(rgb)
        return sig, rgb

    @staticmethod
    def backward(ctx, g_sig, g_rgb):
        xyz, dirs, code, *params = ctx.saved_tensors
        dec = ctx.dec
        dec._pack(dict(zip(ctx.names, params)))
        if g_sig is None:
            g_sig = xyz.new_zeros(xyz.reshape(-1, 3).shape[0])
        g_code, g = dec._backward(xyz, dirs if ctx.has_dirs else None, code, g_sig, g_rgb if ctx.has_dirs else None, ctx.needs_input_grad[4])
        out = [g[n].reshape(p.shape) if ctx.needs_input_grad[5 + k] else None for k, (n, p) in enumerate(zip(ctx.names, params))]
        return (None, None, None, None, g_code.to(code.dtype) if g_code is not None else None, *out)
