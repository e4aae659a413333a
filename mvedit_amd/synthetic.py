"""Seeded synthetic weights in diffusers' state-dict layout (names and shapes of `UNet2DConditionModel` / `ControlNetModel`,
diffusers 0.27.2) for benchmarks and smoke runs -- no checkpoint can be downloaded offline.  Product-side utility: bench.py's
measured path uses this module, never `oracle/` (tests/test_unet.py checks that it agrees with the oracle's own inventory)."""
import math

import torch

CN_EMB = (16, 32, 96, 256)


def _resnet_shapes(p, cin, cout, temb):
    s = {f'{p}.norm1.weight': (cin,), f'{p}.norm1.bias': (cin,),
         f'{p}.conv1.weight': (cout, cin, 3, 3), f'{p}.conv1.bias': (cout,),
         f'{p}.time_emb_proj.weight': (cout, temb), f'{p}.time_emb_proj.bias': (cout,),
         f'{p}.norm2.weight': (cout,), f'{p}.norm2.bias': (cout,),
         f'{p}.conv2.weight': (cout, cout, 3, 3), f'{p}.conv2.bias': (cout,)}
    if cin != cout:
        s[f'{p}.conv_shortcut.weight'] = (cout, cin, 1, 1)
        s[f'{p}.conv_shortcut.bias'] = (cout,)
    return s


def _transformer_shapes(p, c, ctx, layers, linear):
    proj = (c, c) if linear else (c, c, 1, 1)
    s = {f'{p}.norm.weight': (c,), f'{p}.norm.bias': (c,),
         f'{p}.proj_in.weight': proj, f'{p}.proj_in.bias': (c,),
         f'{p}.proj_out.weight': proj, f'{p}.proj_out.bias': (c,)}
    for k in range(layers):
        b = f'{p}.transformer_blocks.{k}'
        for n in ('norm1', 'norm2', 'norm3'):
            s[f'{b}.{n}.weight'] = (c,)
            s[f'{b}.{n}.bias'] = (c,)
        for a, kv in (('attn1', c), ('attn2', ctx)):
            s[f'{b}.{a}.to_q.weight'] = (c, c)
            s[f'{b}.{a}.to_k.weight'] = (c, kv)
            s[f'{b}.{a}.to_v.weight'] = (c, kv)
            s[f'{b}.{a}.to_out.0.weight'] = (c, c)
            s[f'{b}.{a}.to_out.0.bias'] = (c,)
        s[f'{b}.ff.net.0.proj.weight'] = (8 * c, c)
        s[f'{b}.ff.net.0.proj.bias'] = (8 * c,)
        s[f'{b}.ff.net.2.weight'] = (c, 4 * c)
        s[f'{b}.ff.net.2.bias'] = (c,)
    return s


def param_shapes(cfg):
    """Ordered {name: shape} of UNet2DConditionModel(**cfg).state_dict() in diffusers 0.27.2."""
    ch = cfg['block_out_channels']
    L = cfg['layers_per_block']
    temb = ch[0] * 4
    ctx = cfg['cross_attention_dim']
    lin = cfg['use_linear_projection']
    s = {'conv_in.weight': (ch[0], cfg['in_channels'], 3, 3), 'conv_in.bias': (ch[0],),
         'time_embedding.linear_1.weight': (temb, ch[0]), 'time_embedding.linear_1.bias': (temb,),
         'time_embedding.linear_2.weight': (temb, temb), 'time_embedding.linear_2.bias': (temb,)}
    n = len(ch)
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(L):
            s.update(_resnet_shapes(f'down_blocks.{i}.resnets.{j}', cin if j == 0 else cout, cout, temb))
            if cfg['down_attn'][i]:
                s.update(_transformer_shapes(f'down_blocks.{i}.attentions.{j}', cout, ctx, cfg['transformer_layers'][i], lin))
        if i < n - 1:
            s[f'down_blocks.{i}.downsamplers.0.conv.weight'] = (cout, cout, 3, 3)
            s[f'down_blocks.{i}.downsamplers.0.conv.bias'] = (cout,)
        cin = cout
    s.update(_resnet_shapes('mid_block.resnets.0', ch[-1], ch[-1], temb))
    s.update(_transformer_shapes('mid_block.attentions.0', ch[-1], ctx, cfg['transformer_layers'][-1], lin))
    s.update(_resnet_shapes('mid_block.resnets.1', ch[-1], ch[-1], temb))
    rev = list(reversed(ch))
    rev_attn = list(reversed(cfg['down_attn']))
    rev_tl = list(reversed(cfg['transformer_layers']))
    prev = rev[0]
    for i, cout in enumerate(rev):
        cin_blk = rev[min(i + 1, n - 1)]
        for j in range(L + 1):
            skip = cin_blk if j == L else cout
            rin = prev if j == 0 else cout
            s.update(_resnet_shapes(f'up_blocks.{i}.resnets.{j}', rin + skip, cout, temb))
            if rev_attn[i]:
                s.update(_transformer_shapes(f'up_blocks.{i}.attentions.{j}', cout, ctx, rev_tl[i], lin))
        if i < n - 1:
            s[f'up_blocks.{i}.upsamplers.0.conv.weight'] = (cout, cout, 3, 3)
            s[f'up_blocks.{i}.upsamplers.0.conv.bias'] = (cout,)
        prev = cout
    s['conv_norm_out.weight'] = (ch[0],)
    s['conv_norm_out.bias'] = (ch[0],)
    s['conv_out.weight'] = (cfg['out_channels'], ch[0], 3, 3)
    s['conv_out.bias'] = (cfg['out_channels'],)
    return s


def make_state_dict(cfg, seed=1234, dtype=torch.float32):
    """Seeded random weights (no checkpoint is available offline).  Weights ~ U(+-sqrt(3/fan_in)) (unit
    gain, so every branch contributes at the scale of the residual stream and wiring errors are visible),
    biases ~ 0.1 N(0,1), norm scales 1 + 0.1 N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith('.bias'):
            t = 0.1 * torch.randn(shape, generator=g)
        elif '.norm' in name or name.startswith('conv_norm_out'):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = math.prod(shape[1:])
            bound = math.sqrt(3.0 / fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        sd[name] = t.to(dtype)
    return sd


def controlnet_param_shapes(cfg, cond_channels=3):
    full = param_shapes(cfg)
    s = {k: v for k, v in full.items() if not (k.startswith('up_blocks.') or k.startswith('conv_norm_out') or k.startswith('conv_out'))}
    ch = cfg['block_out_channels']
    e = 'controlnet_cond_embedding.'
    s[e + 'conv_in.weight'] = (CN_EMB[0], cond_channels, 3, 3)
    s[e + 'conv_in.bias'] = (CN_EMB[0],)
    for k in range(6):
        ci, co = CN_EMB[k // 2], CN_EMB[(k + 1) // 2]
        s[f'{e}blocks.{k}.weight'] = (co, ci, 3, 3)
        s[f'{e}blocks.{k}.bias'] = (co,)
    s[e + 'conv_out.weight'] = (ch[0], CN_EMB[3], 3, 3)
    s[e + 'conv_out.bias'] = (ch[0],)
    outs = [ch[0]]
    for i, c in enumerate(ch):
        outs += [c] * cfg['layers_per_block']
        if i < len(ch) - 1:
            outs.append(c)
    for k, c in enumerate(outs):
        s[f'controlnet_down_blocks.{k}.weight'] = (c, c, 1, 1)
        s[f'controlnet_down_blocks.{k}.bias'] = (c,)
    s['controlnet_mid_block.weight'] = (ch[-1], ch[-1], 1, 1)
    s['controlnet_mid_block.bias'] = (ch[-1],)
    return s


def make_controlnet_state_dict(cfg, seed=777, dtype=torch.float32, cond_channels=3):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in controlnet_param_shapes(cfg, cond_channels).items():
        if name.endswith('.bias'):
            t = 0.1 * torch.randn(shape, generator=g)
        elif '.norm' in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(3.0 / math.prod(shape[1:]))
        sd[name] = t.to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------------------------------------
# AutoencoderKL (diffusers 0.27.2 state-dict names); tests/test_vae.py checks the inventory against the oracle's
# ---------------------------------------------------------------------------------------------------------------------------
def vae_param_shapes(cfg):
    ch, L, lat = tuple(cfg['block_out_channels']), cfg['layers_per_block'], cfg['latent_channels']
    n = len(ch)

    def resnet(p, cin, cout):
        s = {k: v for k, v in _resnet_shapes(p, cin, cout, 1).items() if '.time_emb_proj.' not in k}
        return s

    def mid(p, c):
        s = resnet(f'{p}.resnets.0', c, c)
        a = f'{p}.attentions.0'
        s[f'{a}.group_norm.weight'] = (c,)
        s[f'{a}.group_norm.bias'] = (c,)
        for nm in ('to_q', 'to_k', 'to_v', 'to_out.0'):
            s[f'{a}.{nm}.weight'] = (c, c)
            s[f'{a}.{nm}.bias'] = (c,)
        s.update(resnet(f'{p}.resnets.1', c, c))
        return s

    s = {'encoder.conv_in.weight': (ch[0], cfg['in_channels'], 3, 3), 'encoder.conv_in.bias': (ch[0],)}
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(L):
            s.update(resnet(f'encoder.down_blocks.{i}.resnets.{j}', cin if j == 0 else cout, cout))
        if i < n - 1:
            s[f'encoder.down_blocks.{i}.downsamplers.0.conv.weight'] = (cout, cout, 3, 3)
            s[f'encoder.down_blocks.{i}.downsamplers.0.conv.bias'] = (cout,)
        cin = cout
    s.update(mid('encoder.mid_block', ch[-1]))
    s.update({'encoder.conv_norm_out.weight': (ch[-1],), 'encoder.conv_norm_out.bias': (ch[-1],),
              'encoder.conv_out.weight': (2 * lat, ch[-1], 3, 3), 'encoder.conv_out.bias': (2 * lat,),
              'decoder.conv_in.weight': (ch[-1], lat, 3, 3), 'decoder.conv_in.bias': (ch[-1],)})
    s.update(mid('decoder.mid_block', ch[-1]))
    cin = ch[-1]
    for i, cout in enumerate(ch[::-1]):
        for j in range(L + 1):
            s.update(resnet(f'decoder.up_blocks.{i}.resnets.{j}', cin if j == 0 else cout, cout))
        if i < n - 1:
            s[f'decoder.up_blocks.{i}.upsamplers.0.conv.weight'] = (cout, cout, 3, 3)
            s[f'decoder.up_blocks.{i}.upsamplers.0.conv.bias'] = (cout,)
        cin = cout
    s.update({'decoder.conv_norm_out.weight': (ch[0],), 'decoder.conv_norm_out.bias': (ch[0],),
              'decoder.conv_out.weight': (cfg['out_channels'], ch[0], 3, 3), 'decoder.conv_out.bias': (cfg['out_channels'],),
              'quant_conv.weight': (2 * lat, 2 * lat, 1, 1), 'quant_conv.bias': (2 * lat,),
              'post_quant_conv.weight': (lat, lat, 1, 1), 'post_quant_conv.bias': (lat,)})
    return s


def make_vae_state_dict(cfg, seed=4321, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in vae_param_shapes(cfg).items():
        if name.endswith('.bias'):
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(math.prod(shape[1:]))
        sd[name] = t.to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------------------------------------
# SRVGGNetCompact (lib/models/decoders/image_space_ss.py state-dict names); tests/test_image_enhancer.py checks the inventory
# ---------------------------------------------------------------------------------------------------------------------------
def srvgg_param_shapes(num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=32, upscale=4):
    s = {'body.0.weight': (num_feat, num_in_ch, 3, 3), 'body.0.bias': (num_feat,), 'body.1.weight': (num_feat,)}
    for k in range(1, num_conv + 1):
        s[f'body.{2 * k}.weight'] = (num_feat, num_feat, 3, 3)
        s[f'body.{2 * k}.bias'] = (num_feat,)
        s[f'body.{2 * k + 1}.weight'] = (num_feat,)
    last = 2 * (num_conv + 1)
    s[f'body.{last}.weight'] = (num_out_ch * upscale ** 2, num_feat, 3, 3)
    s[f'body.{last}.bias'] = (num_out_ch * upscale ** 2,)
    return s


def make_srvgg_state_dict(seed=99, dtype=torch.float32, **kw):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in srvgg_param_shapes(**kw).items():
        if name.endswith('.bias'):
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = 0.25 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) * math.sqrt(1.6 / (shape[1] * 9))
        sd[name] = t.to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------------------------------------------------
# TRACER-B7 segmentor (lib/models/segmentors/tracer_b7.py): the reference modules' parameter names (num_batches_tracked left out)
# ---------------------------------------------------------------------------------------------------------------------------------------
def tracer_param_shapes():
    from .segmentor import block_table, _round_filters, RFB_CH, FEAT_CH
    s = {}

    def bn(name, c):
        for p in ('weight', 'bias', 'running_mean', 'running_var'):
            s[f'{name}.{p}'] = (c,)

    def basic(name, cin, cout, k):
        kh, kw = (k, k) if isinstance(k, int) else k
        s[f'{name}.conv.weight'] = (cout, cin, kh, kw)
        bn(f'{name}.bn', cout)

    _, blocks = block_table()
    s['encoder._conv_stem.weight'] = (_round_filters(32), 3, 3, 3)
    bn('encoder._bn0', _round_filters(32))
    for n, (k, st, e, cin, cout, se, pad) in enumerate(blocks):
        b, mid = f'encoder._blocks.{n}', cin * e
        if e != 1:
            s[f'{b}._expand_conv.weight'] = (mid, cin, 1, 1)
            bn(f'{b}._bn0', mid)
        s[f'{b}._depthwise_conv.weight'] = (mid, 1, k, k)
        bn(f'{b}._bn1', mid)
        s[f'{b}._se_reduce.weight'] = (se, mid, 1, 1)
        s[f'{b}._se_reduce.bias'] = (se,)
        s[f'{b}._se_expand.weight'] = (mid, se, 1, 1)
        s[f'{b}._se_expand.bias'] = (mid,)
        s[f'{b}._project_conv.weight'] = (cout, mid, 1, 1)
        bn(f'{b}._bn2', cout)
    for name, cin, c in (('rfb2', FEAT_CH[1], RFB_CH[0]), ('rfb3', FEAT_CH[2], RFB_CH[1]), ('rfb4', FEAT_CH[3], RFB_CH[2])):
        basic(f'{name}.branch0.0', cin, c, 1)
        for br, kk in ((1, 3), (2, 5), (3, 7)):
            basic(f'{name}.branch{br}.0', cin, c, 1)
            basic(f'{name}.branch{br}.1', c, c, (1, kk))
            basic(f'{name}.branch{br}.2', c, c, (kk, 1))
            basic(f'{name}.branch{br}.3', c, c, 3)
        basic(f'{name}.conv_cat', 4 * c, c, 3)
        basic(f'{name}.conv_res', cin, c, 1)
    c0, c1, c2 = RFB_CH
    for n, ci, co in (('conv_upsample1', c2, c1), ('conv_upsample2', c2, c0), ('conv_upsample3', c1, c0), ('conv_upsample4', c2, c2),
                      ('conv_upsample5', c2 + c1, c2 + c1), ('conv_concat2', c2 + c1, c2 + c1), ('conv_concat3', c0 + c1 + c2, c0 + c1 + c2)):
        basic(f'agg.{n}', ci, co, 3)
    ct = c0 + c1 + c2
    bn('agg.UAM.bn', ct)
    bn('agg.UAM.norm.0', ct)
    for n in ('channel_q', 'channel_k', 'channel_v', 'fc'):
        s[f'agg.UAM.{n}.weight'] = (ct, ct, 1, 1)
    for n in ('spatial_q', 'spatial_k', 'spatial_v'):
        s[f'agg.UAM.{n}.weight'] = (1, ct, 1, 1)
    for name, ch in (('ObjectAttention2', FEAT_CH[1]), ('ObjectAttention1', FEAT_CH[0])):
        h = ch // 2
        s[f'{name}.DWSConv.DWConv.weight'] = (ch, 1, 3, 3)
        bn(f'{name}.DWSConv.bn', ch)
        s[f'{name}.DWSConv.PWConv.weight'] = (h, ch, 1, 1)
        bn(f'{name}.DWSConv.bn2', h)
        for i, k in ((1, 1), (2, 3), (3, 3), (4, 3)):
            s[f'{name}.DWConv{i}.0.DWConv.weight'] = (h, 1, k, k)
            bn(f'{name}.DWConv{i}.0.bn', h)
            basic(f'{name}.DWConv{i}.1', h, ch // 8, 1)
        basic(f'{name}.conv1', h, 1, 1)
    return s


def make_tracer_state_dict(seed=0, dtype=torch.float32):
    """Seeded stand-in for the Carve/tracer_b7 checkpoint (not reachable offline): variance-preserving convolutions and BatchNorm
    statistics near the identity keep the 55-block encoder's activations in range."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in tracer_param_shapes().items():
        if name.endswith('running_var'):
            t = 1.0 + 0.3 * torch.rand(shape, generator=g)
        elif name.endswith('running_mean'):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith('.bias'):
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) * math.sqrt(1.7 / (shape[1] * shape[2] * shape[3]))
        out[name] = t.to(dtype)
    return out


# lpips.LPIPS(net='vgg') as lib/models/losses/lpips_loss.py:8-42 builds it: torchvision VGG16 `features` convolutions in lpips' five slices, the
# input scaling layer and one non-negative 1x1 `lin` weight per slice
_VGG_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
_VGG_CH = ((3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512))
_VGG_SLICE = (1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5)


def lpips_param_shapes():
    s = {'scaling_layer.shift': (1, 3, 1, 1), 'scaling_layer.scale': (1, 3, 1, 1)}
    for i, (ci, co) in enumerate(_VGG_CH):
        s[f'net.slice{_VGG_SLICE[i]}.{_VGG_IDX[i]}.weight'] = (co, ci, 3, 3)
        s[f'net.slice{_VGG_SLICE[i]}.{_VGG_IDX[i]}.bias'] = (co,)
    for k, c in enumerate((64, 128, 256, 512, 512)):
        s[f'lin{k}.model.1.weight'] = (1, c, 1, 1)
    return s


def make_lpips_state_dict(seed=0, dtype=torch.float32):
    """Seeded stand-in for the lpips / torchvision VGG16 checkpoints (not reachable offline): He-initialised convolutions (activations stay
    O(1) through 13 ReLU layers), non-negative `lin` weights as in the trained model, lpips' published input shift / scale."""
    g = torch.Generator().manual_seed(seed)
    out = {'scaling_layer.shift': torch.tensor((-.030, -.088, -.188)).view(1, 3, 1, 1), 'scaling_layer.scale': torch.tensor((.458, .448, .450)).view(1, 3, 1, 1)}
    for name, shape in lpips_param_shapes().items():
        if name in out:
            continue
        if name.endswith('.bias'):
            t = 0.05 * torch.randn(shape, generator=g)
        elif name.startswith('lin'):
            t = torch.rand(shape, generator=g) * 2.0 / shape[1]
        else:
            t = torch.randn(shape, generator=g) * (2.0 / (shape[1] * 9)) ** 0.5
        out[name] = t.to(dtype)
    return out
