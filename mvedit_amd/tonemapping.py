"""Host-side mirror of `lib/models/decoders/tonemapping.py` (class Tonemapping: `smooth_forward`, `lut`, `inverse_lut`, buffers
`lut_x` / `lut_y`) and of the shading expression the pipelines wrap around it for every rendered batch
(lib/pipelines/mvedit_3d_pipeline.py:1372-1384): native single-pass kernels (csrc/shading.hip), no PyTorch fallback for CUDA tensors."""
import torch

from . import _lib


class _LutFn(torch.autograd.Function):
    """Tonemapping.lut / inverse_lut with the gradient torch autograd gives the reference's expressions (mve_tonemap_lut_backward)."""

    @staticmethod
    def forward(ctx, v, lut_x, lut_y, inverse, linear):
        x = v.detach().to(torch.float32).contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.call('mve_tonemap_lut', _lib.ptr(x), x.numel(), _lib.ptr(lut_x), _lib.ptr(lut_y), lut_x.numel(), int(inverse), int(linear),
                      _lib.ptr(out), _lib.stream_ptr(x.device))
        ctx.save_for_backward(x, lut_x, lut_y)
        ctx.mode, ctx.dtype = (int(inverse), int(linear)), v.dtype
        return out.to(v.dtype)

    @staticmethod
    def backward(ctx, g):
        x, lut_x, lut_y = ctx.saved_tensors
        g = g.detach().to(torch.float32).contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.call('mve_tonemap_lut_backward', _lib.ptr(x), _lib.ptr(g), x.numel(), _lib.ptr(lut_x), _lib.ptr(lut_y), lut_x.numel(), ctx.mode[0],
                      ctx.mode[1], _lib.ptr(out), _lib.stream_ptr(x.device))
        return out.to(ctx.dtype), None, None, None, None


class Tonemapping:
    def __init__(self, exposure=0.0, contrast=0.953, bias=0.088, sigmoid_gain=0.943, log_gain=0.011, lut_logx_min=-9, lut_logx_max=3,
                 lut_steps=16, device='cuda'):
        self.exposure, self.contrast, self.bias, self.sigmoid_gain, self.log_gain = exposure, contrast, bias, sigmoid_gain, log_gain
        # 16 knots: host arithmetic, the same torch expressions as the reference's constructor (tonemapping.py:22-31)
        self.lut_x = torch.linspace(lut_logx_min, lut_logx_max, lut_steps)
        self.lut_y = self._smooth(self.lut_x)
        self.to(device)

    def to(self, device=None, **unused):
        if device is not None:
            self.lut_x, self.lut_y = self.lut_x.to(device).contiguous(), self.lut_y.to(device).contiguous()
        return self

    def _smooth(self, x):
        x = (x + self.exposure) * self.contrast
        return x.sigmoid() * self.sigmoid_gain + x * self.log_gain + self.bias

    def smooth_forward(self, x, input_mode='log'):
        """The analytic curve (only evaluated at the knots by the pipelines; kept for API parity, plain torch on 16 values)."""
        assert input_mode in ['log', 'linear']
        if input_mode == 'linear':
            x = x.clamp(min=1e-6).log2()
        return self._smooth(x)

    def _run(self, v, inverse, linear):
        assert v.is_cuda, 'native path: CUDA tensors only'
        return _LutFn.apply(v, self.lut_x, self.lut_y, inverse, linear)       # differentiable w.r.t. v, like the reference's expressions

    def lut(self, x, input_mode='log'):
        assert input_mode in ['log', 'linear']
        return self._run(x, False, input_mode == 'linear')

    def inverse_lut(self, y, output_mode='log'):
        assert output_mode in ['log', 'linear']
        return self._run(y, True, output_mode == 'linear')


def shade_views(rgba, normal_fg, cam_lights, ambient_light, bg_color, tonemapping=None):
    """rgba [..., b, S, S, 4], normal_fg [..., b, S, S, 3] (as `BaseNeRF.render` returns them), cam_lights [b, 3] ->
    image [..., b, S, S, 3]: the reference's `image_batch` (mvedit_3d_pipeline.py:1372-1384) in one launch."""
    assert rgba.is_cuda and rgba.shape[-1] == 4 and normal_fg.shape[-1] == 3 and rgba.shape[:-1] == normal_fg.shape[:-1]
    if torch.is_grad_enabled() and (rgba.requires_grad or normal_fg.requires_grad):
        raise NotImplementedError('shade_views: render-step shading, forward only (the optimisation loops use recon_loss.nerf_optim_loss)')
    b = cam_lights.shape[0]
    lead = rgba.shape[:-1]
    n = rgba.numel() // 4
    assert n % b == 0 and cam_lights.shape == (b, 3)
    c = rgba.float().contiguous()
    nf = normal_fg.float().contiguous()
    lights = cam_lights.to(device=c.device, dtype=torch.float32).contiguous()
    out = torch.empty(*lead, 3, dtype=torch.float32, device=c.device)
    lx, ly, steps = (tonemapping.lut_x, tonemapping.lut_y, tonemapping.lut_x.numel()) if tonemapping is not None else (None, None, 0)
    with torch.cuda.device(c.device):
        _lib.call('mve_shade_views', _lib.ptr(c), _lib.ptr(nf), _lib.ptr(lights), b, n // b, float(ambient_light), float(bg_color),
                  _lib.ptr(lx), _lib.ptr(ly), steps, _lib.ptr(out), _lib.stream_ptr(c.device))
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# The shading functions the pipelines hand to MeshRenderer.forward (lib/pipelines/mvedit_3d_pipeline.py:410-450), on one fused kernel
# (mve_shade_points) with its backward: they run inside every mesh-optimisation iteration and back-propagate into the decoder (albedo)
# and the mesh (world_normal).
class _ShadePointsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, albedo, normal, lights, ambient, lut_x, lut_y):
        assert albedo.is_cuda and albedo.dim() == 2 and albedo.shape[1] == 3 and normal.shape == albedo.shape and lights.shape == albedo.shape
        a, n, l = (t.detach().to(torch.float32).contiguous() for t in (albedo, normal, lights))
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.call('mve_shade_points', _lib.ptr(a), _lib.ptr(n), _lib.ptr(l), a.shape[0], float(ambient), _lib.ptr(lut_x), _lib.ptr(lut_y),
                      0 if lut_x is None else lut_x.numel(), _lib.ptr(out), None, None, None, _lib.stream_ptr(a.device))
        ctx.save_for_backward(a, n, l, lut_x, lut_y)        # version-checked: an in-place change before backward raises instead of giving wrong gradients
        ctx.ambient, ctx.dtypes = float(ambient), (albedo.dtype, normal.dtype)
        return out.to(albedo.dtype)

    @staticmethod
    def backward(ctx, g):
        a, n, l, lut_x, lut_y = ctx.saved_tensors
        g = g.detach().to(torch.float32).contiguous()
        ga, gn = torch.empty_like(a), torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.call('mve_shade_points', _lib.ptr(a), _lib.ptr(n), _lib.ptr(l), a.shape[0], ctx.ambient, _lib.ptr(lut_x), _lib.ptr(lut_y),
                      0 if lut_x is None else lut_x.numel(), None, _lib.ptr(g), _lib.ptr(ga), _lib.ptr(gn), _lib.stream_ptr(a.device))
        return ga.to(ctx.dtypes[0]), gn.to(ctx.dtypes[1]), None, None, None, None


def shade_points(albedo, world_normal, lights, ambient_light, tonemapping=None):
    """albedo / world_normal / lights [N, 3] -> shaded colour [N, 3]; differentiable w.r.t. albedo and world_normal"""
    lx, ly = (tonemapping.lut_x, tonemapping.lut_y) if tonemapping is not None else (None, None)
    return _ShadePointsFn.apply(albedo, world_normal, lights, ambient_light, lx, ly)


def make_shading_fun(worldspace_point_lights, ambient_light, tonemapping=None):
    """`MVEdit3DPipeline.make_shading_fun` (:410-423): worldspace_point_lights [b, h, w, 3] per pixel; the mesh's own albedo is shaded"""
    def shading_fun(world_pos=None, albedo=None, world_normal=None, fg_mask=None, **kwargs):
        return shade_points(albedo, world_normal, worldspace_point_lights[fg_mask.squeeze(0)], ambient_light, tonemapping)
    return shading_fun


def make_nerf_shading_fun(point_albedo, worldspace_point_lights, ambient_light, tonemapping=None):
    """`make_nerf_shading_fun` (:425-442).  point_albedo(world_pos [N, 3]) -> [N, 3] stands for
    `self.nerf.decoder.point_decode(world_pos[None], None, nerf_code)[1].squeeze(0)`, e.g. `lambda x: dec.point_decode_autograd(x)[1]`."""
    def shading_fun(world_pos=None, albedo=None, world_normal=None, fg_mask=None, **kwargs):
        if len(world_pos) == 0:
            return world_pos if albedo is None else albedo
        return shade_points(point_albedo(world_pos), world_normal, worldspace_point_lights[fg_mask.squeeze(0)], ambient_light, tonemapping)
    return shading_fun


def make_nerf_albedo_shading_fun(point_albedo):
    """`make_nerf_albedo_shading_fun` (:444-450): the decoder's colour at the surface points, unshaded"""
    def shading_fun(world_pos=None, albedo=None, **kwargs):
        if len(world_pos) == 0:
            return world_pos if albedo is None else albedo
        return point_albedo(world_pos)
    return shading_fun
