"""Host-side mirror of the reference's `lib/ops/shencoder` (sphere_harmonics.py: `sh_encode`, `SHEncoder`) on the native kernel
(csrc/sh.hip): same names, arguments and autograd behaviour (gradient w.r.t. the inputs only when `calc_grad_inputs`).  CUDA tensors only."""
import torch

from . import _lib


class _sh_encoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        assert inputs.is_cuda and inputs.dim() == 2 and inputs.shape[1] == 3, 'inputs: CUDA tensor [B, 3]'
        x = inputs.detach().to(torch.float32).contiguous()                 # custom_fwd(cast_inputs=torch.float32) (sphere_harmonics.py:17)
        B = x.shape[0]
        out = torch.empty(B, degree ** 2, dtype=torch.float32, device=x.device)
        dy_dx = torch.empty(B, 3 * degree ** 2, dtype=torch.float32, device=x.device) if calc_grad_inputs else None
        with torch.cuda.device(x.device):
            _lib.call('mve_sh_encode', _lib.ptr(x), B, int(degree), _lib.ptr(out), _lib.ptr(dy_dx), _lib.stream_ptr(x.device))
        ctx.dy_dx, ctx.dims = dy_dx, (B, int(degree))
        return out

    @staticmethod
    def backward(ctx, grad):
        if ctx.dy_dx is None:
            return None, None, None
        B, degree = ctx.dims
        g = grad.detach().to(torch.float32).contiguous()
        gi = torch.empty(B, 3, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call('mve_sh_encode_backward', _lib.ptr(g), _lib.ptr(ctx.dy_dx), B, degree, _lib.ptr(gi), _lib.stream_ptr(g.device))
        return gi, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(torch.nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2
        assert self.input_dim == 3, 'SH encoder only support input dim == 3'
        assert 0 < self.degree <= 8, 'SH encoder only supports degree in [1, 8]'

    def __repr__(self):
        return f'SHEncoder: input_dim={self.input_dim} degree={self.degree}'

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        return sh_encode(inputs, self.degree, inputs.requires_grad).reshape(prefix + [self.output_dim])
