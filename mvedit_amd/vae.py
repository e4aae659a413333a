"""Host-side mirror of the AutoencoderKL seam of the reference's denoise loop on top of the native executor (csrc/unet.hip in
VAE mode): the calls it replaces are

    self.vae.decode(x0 / self.vae.config.scaling_factor, return_dict=False)[0]      lib/pipelines/mvedit_3d_pipeline.py:1258-1262,
                                                                                    lib/pipelines/adapter3d_mixin.py:327-338
    self.vae.encode(images * 2 - 1, return_dict=False)[0].mean                      lib/pipelines/mvedit_3d_pipeline.py:1439-1443
    self.vae.encode(images * 2 - 1).latent_dist.sample()                            lib/pipelines/mvedit_3d_pipeline.py:1118-1120

(diffusers==0.27.2 AutoencoderKL; state-dict names unchanged).  Both halves are separate native engines that share the UNet's
conv / GroupNorm / GEMM kernels; there is no PyTorch fallback."""
import ctypes
from types import SimpleNamespace

import torch

from . import _lib
from .ops import dt as _dt

SD_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                     norm_num_groups=32, scaling_factor=0.18215)
OP_CLASSES = ('conv', 'linear', 'attention', 'norm', 'other')


class DiagonalGaussianDistribution:
    """diffusers' posterior object as the reference uses it: `.mean`, `.sample()`, `.mode()` (logvar clamped to [-30, 20])."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class _Half:
    """One native engine: the decoder (+ post_quant_conv) or the encoder (+ quant_conv)."""

    def __init__(self, half, cfg, dtype, device):
        self.half, self.cfg, self.dtype, self.device = half, cfg, dtype, torch.device(device)
        ch = cfg['block_out_channels']
        lat = cfg['latent_channels']
        cin, cout = (lat, cfg['out_channels']) if half == 1 else (cfg['in_channels'], 2 * lat)
        self.cin, self.cout, self.factor = cin, cout, 2 ** (len(ch) - 1)
        self._h = ctypes.c_void_p()
        _lib.call('mve_vae_create', ctypes.byref(self._h), _dt(dtype), half, cin, cout, len(ch), (ctypes.c_int * len(ch))(*[int(c) for c in ch]),
                  int(cfg['layers_per_block']), int(cfg['norm_num_groups']), 1e-6)
        self._ws = None

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _lib.raw('mve_unet_destroy')(h)
            except Exception:
                pass
            self._h = None

    def load(self, state_dict):
        own = ('decoder.', 'post_quant_conv.') if self.half == 1 else ('encoder.', 'quant_conv.')
        with torch.cuda.device(self.device):
            s = _lib.stream_ptr(self.device)
            for name, t in state_dict.items():
                if not name.startswith(own):
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
        if missing:
            raise KeyError(f'{missing} VAE parameters missing from the state dict (first: {own[0]}{buf.value.decode()})')

    def max_batch(self, H, W):
        """Largest batch whose widest image-resolution activation stays below 2^31 elements (32-bit indexing in the conv kernels)."""
        ch = self.cfg['block_out_channels']
        wide = max(ch[0], ch[1]) if len(ch) > 1 else ch[0]
        pix = H * W * (self.factor ** 2 if self.half == 1 else 1)
        return max(1, (2 ** 31 - 1) // (pix * wide))

    def plan(self, B, H, W, io_dtype):
        ws, n_ops, flops = ctypes.c_size_t(), ctypes.c_int(), (ctypes.c_double * 5)()
        _lib.call('mve_vae_plan', self._h, B, H, W, _dt(io_dtype), ctypes.byref(ws), ctypes.byref(n_ops), flops)
        return dict(workspace_bytes=ws.value, n_ops=n_ops.value, flops=dict(zip(OP_CLASSES, list(flops))))

    def op_table(self):
        out, i = [], 0
        cls, fl, lab = ctypes.c_int(), ctypes.c_double(), ctypes.create_string_buffer(96)
        while True:
            ph = _lib.raw('mve_unet_op_info')(self._h, i, ctypes.byref(cls), ctypes.byref(fl), lab, 96)
            if ph < 0:
                break
            out.append((OP_CLASSES[cls.value], fl.value, lab.value.decode()))
            i += 1
        return out

    def run(self, x, max_batch=None, profile=False):
        assert x.dim() == 4 and x.shape[1] == self.cin, (tuple(x.shape), self.cin)
        io = x.dtype if x.dtype in (torch.float32, torch.float16, torch.bfloat16) else torch.float32
        x = x.to(device=self.device, dtype=io).contiguous()
        B, _, H, W = x.shape
        if self.half == 1:
            Ho, Wo = H * self.factor, W * self.factor
        else:
            assert H % self.factor == 0 and W % self.factor == 0, f'image size must be divisible by {self.factor}'
            Ho, Wo = H // self.factor, W // self.factor
        out = torch.empty(B, self.cout, Ho, Wo, dtype=io, device=self.device)
        step = min(B, self.max_batch(H, W), max_batch or B)
        prof = []
        with torch.cuda.device(self.device):
            for b0 in range(0, B, step):       # the reference itself decodes `diff_bs` views at a time (mvedit_3d_pipeline.py:1259)
                nb = min(step, B - b0)
                info = self.plan(nb, H, W, io)
                if self._ws is None or self._ws.numel() < info['workspace_bytes']:
                    self._ws = None
                    self._ws = torch.empty(info['workspace_bytes'], dtype=torch.uint8, device=self.device)
                op_ms = (ctypes.c_float * info['n_ops'])() if profile else None
                _lib.call('mve_vae_forward', self._h, _lib.ptr(x[b0:b0 + nb]), _dt(io), nb, H, W, _lib.ptr(out[b0:b0 + nb]), _lib.ptr(self._ws),
                          self._ws.numel(), op_ms, _lib.stream_ptr(self.device))
                if profile:
                    prof.append([(c, lab, fl, m) for (c, fl, lab), m in zip(self.op_table(), list(op_ms))])
        return (out, prof) if profile else out


class AutoencoderKLEngine:
    """`vae` of the reference's pipelines: `.config.scaling_factor`, `.decode(z, return_dict=False)[0]`,
    `.encode(x).latent_dist` / `.encode(x, return_dict=False)[0]`."""

    def __init__(self, config=None, dtype=torch.float16, device='cuda', max_batch=None):
        self.cfg = dict(config or SD_VAE_CONFIG)
        assert dtype in (torch.float16, torch.bfloat16)
        self.dtype, self.device, self.max_batch = dtype, torch.device(device), max_batch
        self.config = SimpleNamespace(**self.cfg)
        self.decoder = _Half(1, self.cfg, dtype, device)
        self.encoder = _Half(2, self.cfg, dtype, device)

    @classmethod
    def from_state_dict(cls, state_dict, config=None, dtype=torch.float16, device='cuda', max_batch=None):
        eng = cls(config, dtype, device, max_batch)
        eng.load_state_dict(state_dict)
        return eng

    def load_state_dict(self, state_dict, strict=True):
        self.decoder.load(state_dict)
        self.encoder.load(state_dict)
        return self

    def decode(self, z, return_dict=True, generator=None):
        sample = self.decoder.run(z, self.max_batch)
        return SimpleNamespace(sample=sample) if return_dict else (sample,)

    def encode(self, x, return_dict=True):
        posterior = DiagonalGaussianDistribution(self.encoder.run(x, self.max_batch))
        return SimpleNamespace(latent_dist=posterior) if return_dict else (posterior,)
