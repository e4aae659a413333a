"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Functional restatement of SRVGGNetCompact.forward (lib/models/decoders/image_space_ss.py:63-70) over a state dict with the
module's own names (`body.<i>.weight|bias`; PReLU layers hold a per-channel `weight`), torch fp32 on the CPU.

PINNED: tests/golden/srvgg_ref.npz holds the output of the REFERENCE class itself (its definition is executed from
/root/reference by tests/golden/make_srvgg_golden.py; the mmgen registry decorator and the mmcv checkpoint import, both unused by
forward, are dropped) on seeded weights and inputs; tests/test_image_enhancer.py checks this restatement against it."""
import torch
import torch.nn.functional as F


def param_shapes(num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=16, upscale=4):
    s = {'body.0.weight': (num_feat, num_in_ch, 3, 3), 'body.0.bias': (num_feat,), 'body.1.weight': (num_feat,)}
    for k in range(1, num_conv + 1):
        s[f'body.{2 * k}.weight'] = (num_feat, num_feat, 3, 3)
        s[f'body.{2 * k}.bias'] = (num_feat,)
        s[f'body.{2 * k + 1}.weight'] = (num_feat,)
    last = 2 * (num_conv + 1)
    s[f'body.{last}.weight'] = (num_out_ch * upscale ** 2, num_feat, 3, 3)
    s[f'body.{last}.bias'] = (num_out_ch * upscale ** 2,)
    return s


def random_params(seed=0, **kw):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(**kw).items():
        if name.endswith('.bias'):
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = 0.25 + 0.1 * torch.randn(shape, generator=g)          # PReLU slopes (torch initialises them to 0.25)
        else:
            t = torch.randn(shape, generator=g) * (1.6 / (shape[1] * 9)) ** 0.5
        out[name] = t
    return out


def forward(sd, x, upscale=4, q=None):
    """q: optional rounding applied to every layer output (emulates the half-precision module the pipelines run)."""
    q = q or (lambda t: t)
    sd = {k: v.float() for k, v in sd.items()}
    n = max(int(k.split('.')[1]) for k in sd) + 1
    out = q(x.float())
    for i in range(n):
        w = sd[f'body.{i}.weight']
        out = q(F.conv2d(out, w, sd[f'body.{i}.bias'], padding=1) if w.dim() == 4 else F.prelu(out, w))
    return q(F.pixel_shuffle(out, upscale) + F.interpolate(q(x.float()), scale_factor=float(upscale), mode='nearest'))
