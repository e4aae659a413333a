"""Host-side mirror of the ControlNet seam (diffusers ControlNetModel / MultiControlNetModel as the reference calls them,
lib/pipelines/adapter3d_mixin.py:101-116, :173-186, :279-287) on top of the native executor in ControlNet mode (csrc/unet.hip).

    down_block_res_samples, mid_block_res_sample = controlnet(
        sample, t, encoder_hidden_states=..., controlnet_cond=[img, depth], conditioning_scale=[w0, w1],
        guess_mode=False, added_cond_kwargs=None, return_dict=False)

The returned tensors have the reference's logical shape [B, C, h, w] but live in channels-last memory in the engine dtype:
`UNet2DConditionEngine` recognises that and consumes them without a layout conversion; everything else treats them as ordinary
tensors.  MultiControlNetEngine sums the nets' outputs natively (each net accumulates into the same buffers)."""
import ctypes

import torch

from . import _lib
from .ops import dt as _dt
from .unet import UNet2DConditionEngine


class ControlNetEngine(UNet2DConditionEngine):
    def __init__(self, config=None, dtype=torch.float16, device='cuda', conditioning_channels=3, channels_last=True):
        # channels_last=False returns NCHW-contiguous copies (for callers that `.view` the residuals, as the reference's own
        # mixin does when it zero-pads them for paired latents, adapter3d_mixin.py:110-116)
        self.channels_last = channels_last
        # deliberately not calling the UNet constructor: the native object is created in ControlNet mode
        from .unet import SD15_CONFIG
        self.cfg = dict(config or SD15_CONFIG)
        assert dtype in (torch.float16, torch.bfloat16)
        self.dtype = dtype
        self.device = torch.device(device)
        c = self.cfg
        n = len(c['block_out_channels'])
        arr = lambda xs: (ctypes.c_int * n)(*[int(x) for x in xs])
        self._h = ctypes.c_void_p()
        _lib.call('mve_controlnet_create', ctypes.byref(self._h), _dt(dtype), c['in_channels'], int(conditioning_channels), n,
                  arr(c['block_out_channels']), c['layers_per_block'], arr(c['down_attn']), arr(c['num_heads']),
                  arr(c['transformer_layers']), c['cross_attention_dim'], c['norm_num_groups'], float(c['norm_eps']),
                  int(c['use_linear_projection']))
        self._ws = None
        self._ip = (0, 1.0)
        self._ref_keep = None

    def output_shapes(self, B, H, W):
        ch = self.cfg['block_out_channels']
        shapes = [(ch[0], H, W)]
        h, w = H, W
        for i, c in enumerate(ch):
            shapes += [(c, h, w)] * self.cfg['layers_per_block']
            if i + 1 < len(ch):
                h, w = h // 2, w // 2
                shapes.append((c, h, w))
        return shapes, (ch[-1], h, w)

    def new_outputs(self, B, H, W):
        """Zero-copy residual buffers: logical NCHW, channels-last memory, engine dtype."""
        shapes, mid = self.output_shapes(B, H, W)
        mk = lambda s: torch.empty(B, s[1], s[2], s[0], dtype=self.dtype, device=self.device).permute(0, 3, 1, 2)
        return [mk(s) for s in shapes], mk(mid)

    shares_cond = True        # run() takes B / R conditioning images for a batch of B (pipelines/adapter3d_mixin.py: the CFG halves share theirs)

    def run(self, sample, timestep, encoder_hidden_states, cond, scale, down, mid, accumulate, profile=False):
        B, _, H, W = sample.shape
        io = encoder_hidden_states.dtype
        ctx = encoder_hidden_states.to(self.device).contiguous()
        sample = sample.to(device=self.device, dtype=io).contiguous()
        cond = cond.to(device=self.device, dtype=io).contiguous()
        assert cond.shape[2] == 8 * H and cond.shape[3] == 8 * W, 'controlnet_cond must be 8x the latent size'
        # fewer conditioning images than batch items: item b uses image b mod len(cond) (the two halves of a CFG batch share their control
        # images, mvedit_3d_pipeline.py:1232) and the conditioning embedding runs once per image -- bit-identical to repeating the images
        assert B % cond.shape[0] == 0, f'{cond.shape[0]} conditioning images for a batch of {B}'
        if cond.shape[0] < B and self.residual_pair:
            # the plan shares one embedding among R batch items through the conv_in epilogue's residual slot, which the pair mode's conv_in
            # (output = a stream pair) does not have: repeat the images instead (same values, R embeddings)
            cond = cond.repeat(B // cond.shape[0], 1, 1, 1)
        _lib.raw('mve_controlnet_set_cond_repeat')(self._h, B // cond.shape[0])
        t = torch.as_tensor(timestep, dtype=torch.float32, device=self.device).reshape(-1)
        t = t.expand(B).contiguous() if t.numel() == 1 else t.contiguous()
        info = self.plan(B, H, W, ctx.shape[1], 1, False, io)
        ws = self._workspace(info['workspace_bytes'])
        outs = list(down) + [mid]
        ptrs = (ctypes.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        with torch.cuda.device(self.device):
            op_ms = (ctypes.c_float * info['n_ops'])() if profile else None
            _lib.call('mve_controlnet_forward', self._h, _lib.ptr(sample), _dt(io), _lib.ptr(t), _lib.ptr(ctx), _lib.ptr(cond), B, H, W,
                      ctx.shape[1], float(scale), int(bool(accumulate)), ptrs, _lib.ptr(ws), ws.numel(), op_ms,
                      _lib.stream_ptr(self.device))
        if profile:
            return [(c, lab, fl, m) for (ph, c, fl, lab), m in zip(self.op_table(), list(op_ms))]

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0, guess_mode=False,
                 added_cond_kwargs=None, return_dict=False, **unused):
        assert not guess_mode, 'guess_mode is never enabled on this path (adapter3d_mixin.py:107)'
        if added_cond_kwargs:
            raise NotImplementedError('added_cond_kwargs (SDXL) has no reference implementation in MVEdit (SURVEY.md F9)')
        B, _, H, W = sample.shape
        down, mid = self.new_outputs(B, H, W)
        self.run(sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, down, mid, False)
        if not self.channels_last:
            return [d.contiguous() for d in down], mid.contiguous()
        return down, mid

    forward = __call__


class MultiControlNetEngine:
    """MultiControlNetModel: `.nets`, called with lists of conditioning images / scales; the outputs are summed."""

    def __init__(self, nets):
        self.nets = list(nets)

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode=False,
                 added_cond_kwargs=None, return_dict=False, **unused):
        assert len(controlnet_cond) == len(conditioning_scale) == len(self.nets)
        B, _, H, W = sample.shape
        down, mid = self.nets[0].new_outputs(B, H, W)
        for i, (net, cond, scale) in enumerate(zip(self.nets, controlnet_cond, conditioning_scale)):
            net.run(sample, timestep, encoder_hidden_states, cond, scale, down, mid, accumulate=i > 0)
        if not getattr(self.nets[0], 'channels_last', True):
            return [d.contiguous() for d in down], mid.contiguous()
        return down, mid
