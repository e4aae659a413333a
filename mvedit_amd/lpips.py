"""Host-side mirror of the reference's patch loss, `LPIPSLoss` / `lpips.LPIPS(net='vgg')` (lib/models/losses/lpips_loss.py:8-42), on the
native executor (csrc/unet.hip in LPIPS mode, csrc/lpips.hip): forward and backward w.r.t. the prediction as one
`torch.autograd.Function`, so the reference's optimisation loop keeps calling `loss.backward()`.

    lp = LPIPSEngine.from_state_dict(lpips.LPIPS(net='vgg').state_dict(), dtype=torch.bfloat16)
    loss = lp(pred, target)            # [B], differentiable w.r.t. pred; pred / target [B, 3, H, W] in [0, 1]"""
import ctypes

import torch

from . import _lib
from .ops import dt as _dt


class _LPIPSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, eng):
        io = pred.dtype if pred.dtype in (torch.float32, torch.float16, torch.bfloat16) else torch.float32
        p = pred.detach().to(device=eng.device, dtype=io).contiguous()
        t = target.detach().to(device=eng.device, dtype=io).contiguous()
        B, _, H, W = p.shape
        info = eng.plan(B, H, W, io)
        ws = torch.empty(info['workspace_bytes'], dtype=torch.uint8, device=eng.device)     # owned by this call: backward reads it
        loss = torch.empty(B, dtype=torch.float32, device=eng.device)
        with torch.cuda.device(eng.device):
            _lib.call('mve_lpips_forward', eng._h, _lib.ptr(p), _lib.ptr(t), _dt(io), B, H, W, _lib.ptr(loss), _lib.ptr(ws), ws.numel(),
                      _lib.stream_ptr(eng.device))
        ctx.save_for_backward(ws)                  # the activations of this call: backward reads them (and overwrites them: one backward only)
        ctx.eng, ctx.shape, ctx.io, ctx.in_dtype, ctx.done = eng, (B, H, W), io, pred.dtype, False
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        eng, (B, H, W) = ctx.eng, ctx.shape
        if ctx.done:
            raise RuntimeError('LPIPSEngine: a second backward through the same forward is not supported (the backward pass overwrites '
                               'the saved activations); run the forward again')
        ws, = ctx.saved_tensors
        ctx.done = True
        g = grad_loss.to(device=eng.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, 3, H, W, dtype=ctx.io, device=eng.device)
        with torch.cuda.device(eng.device):
            _lib.call('mve_lpips_backward', eng._h, _lib.ptr(g), _dt(ctx.io), B, H, W, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                      _lib.stream_ptr(eng.device))
        return out.to(ctx.in_dtype), None, None


class LPIPSEngine:
    def __init__(self, dtype=torch.bfloat16, device='cuda', normalize_inputs=True):
        assert dtype in (torch.float16, torch.bfloat16)       # the reference runs the module in bf16 when the GPU has it (lpips_loss.py:31)
        self.dtype, self.device, self.normalize_inputs = dtype, torch.device(device), normalize_inputs
        self._h = ctypes.c_void_p()
        _lib.call('mve_lpips_create', ctypes.byref(self._h), _dt(dtype), int(normalize_inputs))

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _lib.raw('mve_unet_destroy')(h)
            except Exception:
                pass
            self._h = None

    @classmethod
    def from_state_dict(cls, state_dict, dtype=torch.bfloat16, device='cuda', normalize_inputs=True):
        return cls(dtype, device, normalize_inputs).load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict=True):
        with torch.cuda.device(self.device):
            s = _lib.stream_ptr(self.device)
            for name, t in state_dict.items():
                if not name.startswith(('net.', 'lin', 'scaling_layer.')):
                    continue
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
            raise KeyError(f'{missing} LPIPS parameters missing from the state dict (first: {buf.value.decode()})')
        return self

    def plan(self, B, H, W, io_dtype=torch.float32):
        ws, n_ops, n_fwd, flops = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), (ctypes.c_double * 5)()
        _lib.call('mve_lpips_plan', self._h, B, H, W, _dt(io_dtype), ctypes.byref(ws), ctypes.byref(n_ops), ctypes.byref(n_fwd), flops)
        return dict(workspace_bytes=ws.value, n_ops=n_ops.value, n_forward_ops=n_fwd.value, conv_flops=flops[0])

    def __call__(self, pred, target):
        """-> [B] fp32 on the engine's device; differentiable w.r.t. pred (target is a constant, as in the reference's loss)."""
        assert pred.shape == target.shape and pred.dim() == 4 and pred.shape[1] == 3, tuple(pred.shape)
        return _LPIPSFn.apply(pred, target, self)

    forward = __call__
