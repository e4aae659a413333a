"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Plain-PyTorch (CPU, fp32) restatement of the AutoencoderKL the reference calls inside its denoise loop:

    self.vae.decode(x0 / scaling_factor, return_dict=False)[0]           lib/pipelines/mvedit_3d_pipeline.py:1258-1262,
                                                                        lib/pipelines/adapter3d_mixin.py:327-338
    self.vae.encode(images * 2 - 1, return_dict=False)[0].mean          lib/pipelines/mvedit_3d_pipeline.py:1439-1443
    self.vae.encode(images * 2 - 1).latent_dist.sample()                lib/pipelines/mvedit_3d_pipeline.py:1118-1120, :1131-1133

The class itself is third-party `diffusers==0.27.2` (requirements.txt:13), absent from /root/reference and from this image; it
is restated from that release's published semantics with torch built-ins only (F.conv2d, F.group_norm, F.silu, softmax):
  * Encoder: conv_in 3x3 -> DownEncoderBlock2D x n (layers_per_block ResnetBlock2D without time embedding, eps 1e-6; all but
    the last followed by Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) then conv3x3 stride 2 without padding)
    -> UNetMidBlock2D (resnet, single-head attention, resnet) -> GroupNorm(eps 1e-6) -> SiLU -> conv_out 3x3 (2 * latent);
    quant_conv 1x1; DiagonalGaussianDistribution: mean, logvar = chunk(2), logvar.clamp(-30, 20), std = exp(logvar / 2);
  * Decoder: post_quant_conv 1x1 -> conv_in 3x3 -> mid block -> UpDecoderBlock2D x n over reversed widths (layers_per_block + 1
    resnets; all but the last followed by Upsample2D: nearest x2 then conv3x3) -> GroupNorm -> SiLU -> conv_out 3x3;
  * mid-block Attention (heads = 1, dim_head = C, residual_connection, bias on q/k/v/out, group_norm eps 1e-6):
    x + to_out(softmax(q k^T / sqrt(C)) v) with q, k, v = Linear(GroupNorm(x) as tokens).
State-dict keys are diffusers' own (`encoder.*`, `decoder.*`, `quant_conv.*`, `post_quant_conv.*`).

PARITY UNPINNED: the reference holds no tests or vectors for this call and diffusers is not importable here; the restatement is
pinned only against itself (tests/golden/vae_tiny.npz, written by tests/golden/make_vae_golden.py)."""
import torch
import torch.nn.functional as F

SD_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
              norm_num_groups=32, scaling_factor=0.18215)
# two-level miniature: one downsample, both shortcut kinds (64 -> 128 fused-able; 128 -> 64)
TINY_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(64, 128), layers_per_block=1,
                norm_num_groups=32, scaling_factor=0.18215)
# widths that are not multiples of 64 (separate shortcut GEMM path)
ODD_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 96), layers_per_block=1,
               norm_num_groups=16, scaling_factor=0.18215)
EPS = 1e-6


def _resnet_shapes(p, cin, cout):
    s = {f'{p}.norm1.weight': (cin,), f'{p}.norm1.bias': (cin,), f'{p}.conv1.weight': (cout, cin, 3, 3), f'{p}.conv1.bias': (cout,),
         f'{p}.norm2.weight': (cout,), f'{p}.norm2.bias': (cout,), f'{p}.conv2.weight': (cout, cout, 3, 3), f'{p}.conv2.bias': (cout,)}
    if cin != cout:
        s[f'{p}.conv_shortcut.weight'] = (cout, cin, 1, 1)
        s[f'{p}.conv_shortcut.bias'] = (cout,)
    return s


def _mid_shapes(p, c):
    s = _resnet_shapes(f'{p}.resnets.0', c, c)
    a = f'{p}.attentions.0'
    s[f'{a}.group_norm.weight'] = (c,)
    s[f'{a}.group_norm.bias'] = (c,)
    for n in ('to_q', 'to_k', 'to_v', 'to_out.0'):
        s[f'{a}.{n}.weight'] = (c, c)
        s[f'{a}.{n}.bias'] = (c,)
    s.update(_resnet_shapes(f'{p}.resnets.1', c, c))
    return s


def param_shapes(cfg):
    """Ordered {name: shape} of AutoencoderKL(**cfg).state_dict() in diffusers 0.27.2."""
    ch, L, lat = cfg['block_out_channels'], cfg['layers_per_block'], cfg['latent_channels']
    n = len(ch)
    s = {'encoder.conv_in.weight': (ch[0], cfg['in_channels'], 3, 3), 'encoder.conv_in.bias': (ch[0],)}
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(L):
            s.update(_resnet_shapes(f'encoder.down_blocks.{i}.resnets.{j}', cin if j == 0 else cout, cout))
        if i < n - 1:
            s[f'encoder.down_blocks.{i}.downsamplers.0.conv.weight'] = (cout, cout, 3, 3)
            s[f'encoder.down_blocks.{i}.downsamplers.0.conv.bias'] = (cout,)
        cin = cout
    s.update(_mid_shapes('encoder.mid_block', ch[-1]))
    s['encoder.conv_norm_out.weight'] = (ch[-1],)
    s['encoder.conv_norm_out.bias'] = (ch[-1],)
    s['encoder.conv_out.weight'] = (2 * lat, ch[-1], 3, 3)
    s['encoder.conv_out.bias'] = (2 * lat,)
    s['decoder.conv_in.weight'] = (ch[-1], lat, 3, 3)
    s['decoder.conv_in.bias'] = (ch[-1],)
    s.update(_mid_shapes('decoder.mid_block', ch[-1]))
    rev = ch[::-1]
    cin = rev[0]
    for i, cout in enumerate(rev):
        for j in range(L + 1):
            s.update(_resnet_shapes(f'decoder.up_blocks.{i}.resnets.{j}', cin if j == 0 else cout, cout))
        if i < n - 1:
            s[f'decoder.up_blocks.{i}.upsamplers.0.conv.weight'] = (cout, cout, 3, 3)
            s[f'decoder.up_blocks.{i}.upsamplers.0.conv.bias'] = (cout,)
        cin = cout
    s['decoder.conv_norm_out.weight'] = (ch[0],)
    s['decoder.conv_norm_out.bias'] = (ch[0],)
    s['decoder.conv_out.weight'] = (cfg['out_channels'], ch[0], 3, 3)
    s['decoder.conv_out.bias'] = (cfg['out_channels'],)
    s['quant_conv.weight'] = (2 * lat, 2 * lat, 1, 1)
    s['quant_conv.bias'] = (2 * lat,)
    s['post_quant_conv.weight'] = (lat, lat, 1, 1)
    s['post_quant_conv.bias'] = (lat,)
    return s


def random_params(cfg, seed=0, dtype=torch.float32):
    """Seeded stand-in weights with activations of order one through every block (no checkpoint can be fetched here)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith('.bias'):
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)
            t = torch.randn(shape, generator=g) / fan_in ** 0.5
        out[name] = t.to(dtype)
    return out


def quantizer(dtype):
    """Round-trip through a storage dtype: emulates PyTorch's half / bfloat16 path (fp32 inside an op, every op's output rounded)."""
    return lambda x: x.to(dtype).float()


def _id(x):
    return x


def _gn_silu(sd, p, x, groups, q, silu=True):
    h = q(F.group_norm(x, groups, sd[f'{p}.weight'], sd[f'{p}.bias'], EPS))
    return q(F.silu(h)) if silu else h


def _resnet(sd, p, x, groups, q):
    h = q(F.conv2d(_gn_silu(sd, f'{p}.norm1', x, groups, q), sd[f'{p}.conv1.weight'], sd[f'{p}.conv1.bias'], padding=1))
    h = q(F.conv2d(_gn_silu(sd, f'{p}.norm2', h, groups, q), sd[f'{p}.conv2.weight'], sd[f'{p}.conv2.bias'], padding=1))
    if f'{p}.conv_shortcut.weight' in sd:
        x = q(F.conv2d(x, sd[f'{p}.conv_shortcut.weight'], sd[f'{p}.conv_shortcut.bias']))
    return q(x + h)


def _attention(sd, p, x, groups, q):
    B, C, H, W = x.shape
    t = q(F.group_norm(x.view(B, C, H * W), groups, sd[f'{p}.group_norm.weight'], sd[f'{p}.group_norm.bias'], EPS)).transpose(1, 2)
    qq, k, v = (q(F.linear(t, sd[f'{p}.{n}.weight'].view(C, C), sd[f'{p}.{n}.bias'])) for n in ('to_q', 'to_k', 'to_v'))
    a = q(torch.softmax(qq @ k.transpose(1, 2) / C ** 0.5, dim=-1) @ v)
    a = q(F.linear(a, sd[f'{p}.to_out.0.weight'].view(C, C), sd[f'{p}.to_out.0.bias']))
    return q(x + a.transpose(1, 2).reshape(B, C, H, W))


def _mid(sd, p, x, groups, q):
    x = _resnet(sd, f'{p}.resnets.0', x, groups, q)
    x = _attention(sd, f'{p}.attentions.0', x, groups, q)
    return _resnet(sd, f'{p}.resnets.1', x, groups, q)


def encode_moments(sd, cfg, x, q=None):
    """AutoencoderKL.encode up to the distribution parameters: [B, 2 * latent, H/2^(n-1), W/2^(n-1)] = cat(mean, logvar)."""
    q = q or _id
    sd = {k: v.float() for k, v in sd.items()}
    ch, L, G = cfg['block_out_channels'], cfg['layers_per_block'], cfg['norm_num_groups']
    h = q(F.conv2d(q(x.float()), sd['encoder.conv_in.weight'], sd['encoder.conv_in.bias'], padding=1))
    for i in range(len(ch)):
        for j in range(L):
            h = _resnet(sd, f'encoder.down_blocks.{i}.resnets.{j}', h, G, q)
        if i < len(ch) - 1:
            p = f'encoder.down_blocks.{i}.downsamplers.0.conv'
            h = q(F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f'{p}.weight'], sd[f'{p}.bias'], stride=2))
    h = _mid(sd, 'encoder.mid_block', h, G, q)
    h = _gn_silu(sd, 'encoder.conv_norm_out', h, G, q)
    h = q(F.conv2d(h, sd['encoder.conv_out.weight'], sd['encoder.conv_out.bias'], padding=1))
    return q(F.conv2d(h, sd['quant_conv.weight'], sd['quant_conv.bias']))


def gaussian(moments):
    """DiagonalGaussianDistribution: -> (mean, std)."""
    mean, logvar = moments.chunk(2, dim=1)
    return mean, torch.exp(0.5 * logvar.clamp(-30.0, 20.0))


def decode(sd, cfg, z, q=None):
    """AutoencoderKL.decode(z, return_dict=False)[0]: [B, latent, h, w] -> [B, out, h * 2^(n-1), w * 2^(n-1)]."""
    q = q or _id
    sd = {k: v.float() for k, v in sd.items()}
    ch, L, G = cfg['block_out_channels'], cfg['layers_per_block'], cfg['norm_num_groups']
    h = q(F.conv2d(q(z.float()), sd['post_quant_conv.weight'], sd['post_quant_conv.bias']))
    h = q(F.conv2d(h, sd['decoder.conv_in.weight'], sd['decoder.conv_in.bias'], padding=1))
    h = _mid(sd, 'decoder.mid_block', h, G, q)
    rev = ch[::-1]
    for i in range(len(ch)):
        for j in range(L + 1):
            h = _resnet(sd, f'decoder.up_blocks.{i}.resnets.{j}', h, G, q)
        if i < len(ch) - 1:
            p = f'decoder.up_blocks.{i}.upsamplers.0.conv'
            h = q(F.conv2d(F.interpolate(h, scale_factor=2.0, mode='nearest'), sd[f'{p}.weight'], sd[f'{p}.bias'], padding=1))
    h = _gn_silu(sd, 'decoder.conv_norm_out', h, G, q)
    return q(F.conv2d(h, sd['decoder.conv_out.weight'], sd['decoder.conv_out.bias'], padding=1))
