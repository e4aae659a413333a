"""Host-side mirror of the pipelines' `image_enhancer` (SRVGGNetCompact, lib/models/decoders/image_space_ss.py:8-70; the x4
Real-ESRGAN "general-x4v3" net the runner builds at lib/pipelines/utils.py:212-215 and the loop calls on every batch of views rendered
below 512 x 512, lib/pipelines/mvedit_3d_pipeline.py:1399-1400) on the native executor: same constructor arguments, same
state-dict names (`body.<i>.weight|bias`), `enhancer(images)` -> [B, C, r H, r W]."""
import ctypes

import torch

from . import _lib
from .ops import dt as _dt

OP_CLASSES = ('conv', 'linear', 'attention', 'norm', 'other')


class SRVGGNetCompactEngine:
    def __init__(self, num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=16, upscale=4, act_type='prelu', dtype=torch.float16, device='cuda'):
        assert act_type == 'prelu', 'the pipelines build the PReLU variant (lib/pipelines/utils.py:213)'
        assert dtype in (torch.float16, torch.bfloat16)
        self.num_in_ch, self.num_out_ch, self.num_feat, self.num_conv, self.upscale = num_in_ch, num_out_ch, num_feat, num_conv, upscale
        self.dtype, self.device = dtype, torch.device(device)
        self._h = ctypes.c_void_p()
        _lib.call('mve_srvgg_create', ctypes.byref(self._h), _dt(dtype), num_in_ch, num_out_ch, num_feat, num_conv, upscale)
        self._ws = None

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _lib.raw('mve_unet_destroy')(h)
            except Exception:
                pass
            self._h = None

    def load_state_dict(self, state_dict, strict=True):
        state_dict = state_dict.get('params', state_dict)          # the released checkpoints keep the weights under 'params'
        with torch.cuda.device(self.device):
            s = _lib.stream_ptr(self.device)
            for name, t in state_dict.items():
                t = t.detach()
                if t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
                    t = t.float()
                t = t.to(self.device).contiguous()
                shape = (ctypes.c_longlong * t.dim())(*t.shape)
                _lib.call('mve_unet_load_param', self._h, name.encode(), _lib.ptr(t), _dt(t), t.dim(), shape, s)
            torch.cuda.current_stream(self.device).synchronize()
        buf = ctypes.create_string_buffer(256)
        missing = _lib.raw('mve_unet_missing_params')(self._h, buf, 256)
        if strict and missing:
            raise KeyError(f'{missing} SRVGGNetCompact parameters missing from the state dict (first: {buf.value.decode()})')
        return self

    def plan(self, B, H, W, io_dtype=None):
        ws, n_ops, flops = ctypes.c_size_t(), ctypes.c_int(), (ctypes.c_double * 5)()
        _lib.call('mve_srvgg_plan', self._h, B, H, W, _dt(io_dtype or self.dtype), ctypes.byref(ws), ctypes.byref(n_ops), flops)
        return dict(workspace_bytes=ws.value, n_ops=n_ops.value, flops=dict(zip(OP_CLASSES, list(flops))))

    def max_batch(self, H, W):
        wide = max(self.num_feat, 2 * ((self.num_out_ch * self.upscale ** 2 + 7) // 8 * 8))
        return max(1, (2 ** 31 - 1) // (H * W * wide))

    def __call__(self, x):
        assert x.dim() == 4 and x.shape[1] == self.num_in_ch, tuple(x.shape)
        io = x.dtype if x.dtype in (torch.float32, torch.float16, torch.bfloat16) else torch.float32
        x = x.to(device=self.device, dtype=io).contiguous()
        B, C, H, W = x.shape
        out = torch.empty(B, self.num_out_ch, H * self.upscale, W * self.upscale, dtype=io, device=self.device)
        step = min(B, self.max_batch(H, W))
        with torch.cuda.device(self.device):
            for b0 in range(0, B, step):
                nb = min(step, B - b0)
                info = self.plan(nb, H, W, io)
                if self._ws is None or self._ws.numel() < info['workspace_bytes']:
                    self._ws = None
                    self._ws = torch.empty(info['workspace_bytes'], dtype=torch.uint8, device=self.device)
                _lib.call('mve_srvgg_forward', self._h, _lib.ptr(x[b0:b0 + nb]), _dt(io), nb, H, W, _lib.ptr(out[b0:b0 + nb]), _lib.ptr(self._ws),
                          self._ws.numel(), None, _lib.stream_ptr(self.device))
        return out

    forward = __call__
