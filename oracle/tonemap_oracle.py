"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Restatement (torch fp32, CPU) of
  * Tonemapping.lut / .inverse_lut / the knot tables (lib/models/decoders/tonemapping.py:22-54), and
  * the shading expression of the render step (lib/pipelines/mvedit_3d_pipeline.py:1372-1384; identical at :155-168).

PINNED: tests/golden/tonemap_ref.npz holds outputs of the REFERENCE's Tonemapping class (imported from /root/reference by
tests/golden/make_tonemap_golden.py -- the module only needs torch) for the tables and both methods in both modes, and the shading
expression evaluated with the reference class's methods around the restated inline arithmetic."""
import torch


def tables(exposure=0.0, contrast=0.953, bias=0.088, sigmoid_gain=0.943, log_gain=0.011, lo=-9, hi=3, steps=16):
    x = torch.linspace(lo, hi, steps)
    z = (x + exposure) * contrast
    return x, z.sigmoid() * sigmoid_gain + z * log_gain + bias


def _interp(a, b, v):
    i = torch.bucketize(v, a, right=True).clamp(min=1, max=len(a) - 1)
    t = (v - a[i - 1]) / (a[i] - a[i - 1])
    return b[i - 1] + (b[i] - b[i - 1]) * t


def lut(lx, ly, x, input_mode='log'):
    x = x.float()
    if input_mode == 'linear':
        x = x.clamp(min=1e-6).log2()
    return _interp(lx, ly, x)


def inverse_lut(lx, ly, y, output_mode='log'):
    x = _interp(ly, lx, y.float())
    return torch.exp2(x) if output_mode == 'linear' else x


def shade_views(rgba, normal_fg, cam_lights, ambient_light, bg_color, lx=None, ly=None, lut_fn=None, inv_fn=None):
    """rgba [1, b, S, S, 4], normal_fg [1, b, S, S, 3], cam_lights [b, 3].  lut_fn / inv_fn: optional callables (the reference's
    own methods when the golden file is generated)."""
    n_cv = torch.cat([normal_fg[..., :1] * 2 - 1, -normal_fg[..., 1:3] * 2 + 1], dim=-1)
    shading = ((cam_lights[:, None, None, None, :] @ n_cv[..., :, None]).clamp(min=0) * (1 - ambient_light) + ambient_light).squeeze(-1)
    if lx is None and lut_fn is None:
        return rgba[..., :3] * shading + bg_color * (1 - rgba[..., 3:])
    lut_fn = lut_fn or (lambda v: lut(lx, ly, v))
    inv_fn = inv_fn or (lambda v: inverse_lut(lx, ly, v))
    return lut_fn(inv_fn(rgba[..., :3] / rgba[..., 3:].clamp(min=1e-6)) + shading.clamp(min=1e-6).log2()) * rgba[..., 3:] \
        + bg_color * (1 - rgba[..., 3:])
