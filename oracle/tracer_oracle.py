"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Functional restatement of the reference's foreground segmentor `TracerUniversalB7.forward` (lib/models/segmentors/tracer_b7.py:56-73)
over a state dict with the reference module's own parameter names, torch fp32 on the CPU:
  * EfficientNet-B7 encoder (lib/models/architecture/tracerb7/efficientnet.py:30-150 MBConvBlock, :270-325 EfficientEncoderB7; the TensorFlow
    "SAME" paddings are the STATIC ones the reference computes at construction for a 600-pixel image, effi_utils.py:270-315, whatever the
    actual input size is), features after blocks 10 / 17 / 37 / 54;
  * TRACER decoder (tracer.py:72-97): RFB blocks, aggregation, union attention (att_modules.py:13-246), two object-attention stages
    (att_modules.py:249-291), the three bilinear side outputs and their mean under a sigmoid;
  * the wrapper's preprocessing (Resize(antialias=False) + Normalize), min-pool erosion, resize back and the failure rule.

PINNED: tests/golden/tracer_ref.npz holds the outputs of the REFERENCE modules themselves (executed from /root/reference by
tests/golden/make_tracer_golden.py on seeded weights and inputs); tests/test_segmentor.py checks this restatement against them.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS_ENC = 1e-3          # effi_utils.py:572
BN_EPS_DEC = 1e-5          # nn.BatchNorm2d default (conv_modules.py, att_modules.py)
FEATURE_BLOCKS = (10, 17, 37, 54)
RFB_CH = (32, 64, 128)
FEAT_CH = (48, 80, 224, 640)


def _round_filters(f, width=2.0, divisor=8):
    f *= width
    nf = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if nf < 0.9 * f:
        nf += divisor
    return int(nf)


def _same_pad(size, k, s):
    """(before, after) of Conv2dStaticSamePadding for one axis (effi_utils.py:290-300); also the output size"""
    o = math.ceil(size / s)
    p = max((o - 1) * s + (k - 1) + 1 - size, 0)
    return (p // 2, p - p // 2), o


def block_table(image_size=600):
    """[(kernel, stride, expand, cin, cout, se_channels, (pad_before, pad_after))] of the 55 MBConv blocks of EfficientNet-B7, plus the stem's padding.
    The paddings follow the CONSTRUCTION image size (600), as the reference's static padding modules do."""
    base = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
    stem_pad, size = _same_pad(image_size, 3, 2)
    blocks = []
    for (r, k, s, e, i, o) in base:
        i, o = _round_filters(i), _round_filters(o)
        n = int(math.ceil(3.1 * r))
        for j in range(n):
            cin, st = (i, s) if j == 0 else (o, 1)
            pad, size2 = _same_pad(size, k, st)
            blocks.append((k, st, e, cin, o, max(1, int(cin * 0.25)), pad))
            size = size2
    return stem_pad, blocks


# ---------------------------------------------------------------------------------------------------------------------------------------
# parameter inventory (the reference modules' state-dict names, num_batches_tracked left out)
# ---------------------------------------------------------------------------------------------------------------------------------------
def _bn(s, name, c):
    for p in ('weight', 'bias', 'running_mean', 'running_var'):
        s[f'{name}.{p}'] = (c,)


def _basic(s, name, cin, cout, k):
    kh, kw = (k, k) if isinstance(k, int) else k
    s[f'{name}.conv.weight'] = (cout, cin, kh, kw)
    _bn(s, f'{name}.bn', cout)


def param_shapes():
    s = {}
    _, blocks = block_table()
    s['encoder._conv_stem.weight'] = (_round_filters(32), 3, 3, 3)
    _bn(s, 'encoder._bn0', _round_filters(32))
    for n, (k, st, e, cin, cout, se, pad) in enumerate(blocks):
        b = f'encoder._blocks.{n}'
        mid = cin * e
        if e != 1:
            s[f'{b}._expand_conv.weight'] = (mid, cin, 1, 1)
            _bn(s, f'{b}._bn0', mid)
        s[f'{b}._depthwise_conv.weight'] = (mid, 1, k, k)
        _bn(s, f'{b}._bn1', mid)
        s[f'{b}._se_reduce.weight'] = (se, mid, 1, 1)
        s[f'{b}._se_reduce.bias'] = (se,)
        s[f'{b}._se_expand.weight'] = (mid, se, 1, 1)
        s[f'{b}._se_expand.bias'] = (mid,)
        s[f'{b}._project_conv.weight'] = (cout, mid, 1, 1)
        _bn(s, f'{b}._bn2', cout)
    for name, cin, c in (('rfb2', FEAT_CH[1], RFB_CH[0]), ('rfb3', FEAT_CH[2], RFB_CH[1]), ('rfb4', FEAT_CH[3], RFB_CH[2])):
        _basic(s, f'{name}.branch0.0', cin, c, 1)
        for br, kk in ((1, 3), (2, 5), (3, 7)):
            _basic(s, f'{name}.branch{br}.0', cin, c, 1)
            _basic(s, f'{name}.branch{br}.1', c, c, (1, kk))
            _basic(s, f'{name}.branch{br}.2', c, c, (kk, 1))
            _basic(s, f'{name}.branch{br}.3', c, c, 3)
        _basic(s, f'{name}.conv_cat', 4 * c, c, 3)
        _basic(s, f'{name}.conv_res', cin, c, 1)
    c0, c1, c2 = RFB_CH
    _basic(s, 'agg.conv_upsample1', c2, c1, 3)
    _basic(s, 'agg.conv_upsample2', c2, c0, 3)
    _basic(s, 'agg.conv_upsample3', c1, c0, 3)
    _basic(s, 'agg.conv_upsample4', c2, c2, 3)
    _basic(s, 'agg.conv_upsample5', c2 + c1, c2 + c1, 3)
    _basic(s, 'agg.conv_concat2', c2 + c1, c2 + c1, 3)
    _basic(s, 'agg.conv_concat3', c0 + c1 + c2, c0 + c1 + c2, 3)
    ct = c0 + c1 + c2
    _bn(s, 'agg.UAM.bn', ct)
    _bn(s, 'agg.UAM.norm.0', ct)
    for n in ('channel_q', 'channel_k', 'channel_v', 'fc'):
        s[f'agg.UAM.{n}.weight'] = (ct, ct, 1, 1)
    for n in ('spatial_q', 'spatial_k', 'spatial_v'):
        s[f'agg.UAM.{n}.weight'] = (1, ct, 1, 1)
    for name, ch in (('ObjectAttention2', FEAT_CH[1]), ('ObjectAttention1', FEAT_CH[0])):
        h = ch // 2
        s[f'{name}.DWSConv.DWConv.weight'] = (ch, 1, 3, 3)
        _bn(s, f'{name}.DWSConv.bn', ch)
        s[f'{name}.DWSConv.PWConv.weight'] = (h, ch, 1, 1)
        _bn(s, f'{name}.DWSConv.bn2', h)
        for i, k in ((1, 1), (2, 3), (3, 3), (4, 3)):
            s[f'{name}.DWConv{i}.0.DWConv.weight'] = (h, 1, k, k)
            _bn(s, f'{name}.DWConv{i}.0.bn', h)
            _basic(s, f'{name}.DWConv{i}.1', h, ch // 8, 1)
        _basic(s, f'{name}.conv1', h, 1, 1)
    return s


def random_params(seed=0):
    """Seeded weights that keep the activations of the 55-block encoder in range (variance-preserving convolutions, BatchNorm statistics near
    the identity): a random-weight stand-in for the Carve/tracer_b7 checkpoint, which is not reachable offline."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes().items():
        if name.endswith('running_var'):
            t = 1.0 + 0.3 * torch.rand(shape, generator=g)
        elif name.endswith('running_mean'):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith('.bias'):
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 1:                        # BatchNorm weight
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(1.7 / fan_in)
        out[name] = t
    return out


# ---------------------------------------------------------------------------------------------------------------------------------------
# forward
# ---------------------------------------------------------------------------------------------------------------------------------------
def _bnf(sd, name, x, eps):
    return F.batch_norm(x, sd[f'{name}.running_mean'], sd[f'{name}.running_var'], sd[f'{name}.weight'], sd[f'{name}.bias'], False, 0.0, eps)


def _swish(x):
    return x * torch.sigmoid(x)


def encoder(sd, x, q):
    """EfficientEncoderB7.forward (efficientnet.py:283-325): list of the four feature maps"""
    stem_pad, blocks = block_table()
    x = F.pad(x, (stem_pad[0], stem_pad[1], stem_pad[0], stem_pad[1]))
    x = q(_swish(_bnf(sd, 'encoder._bn0', F.conv2d(x, sd['encoder._conv_stem.weight'], stride=2), BN_EPS_ENC)))
    feats = []
    for n, (k, st, e, cin, cout, se, pad) in enumerate(blocks):
        b = f'encoder._blocks.{n}'
        inp = x
        if e != 1:
            x = q(_swish(_bnf(sd, f'{b}._bn0', F.conv2d(x, sd[f'{b}._expand_conv.weight']), BN_EPS_ENC)))
        x = F.pad(x, (pad[0], pad[1], pad[0], pad[1]))
        x = q(_swish(_bnf(sd, f'{b}._bn1', F.conv2d(x, sd[f'{b}._depthwise_conv.weight'], stride=st, groups=x.shape[1]), BN_EPS_ENC)))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = _swish(F.conv2d(sq, sd[f'{b}._se_reduce.weight'], sd[f'{b}._se_reduce.bias']))
        sq = F.conv2d(sq, sd[f'{b}._se_expand.weight'], sd[f'{b}._se_expand.bias'])
        x = q(torch.sigmoid(sq) * x)
        x = _bnf(sd, f'{b}._bn2', F.conv2d(x, sd[f'{b}._project_conv.weight']), BN_EPS_ENC)
        if st == 1 and cin == cout:
            x = x + inp                              # eval mode: drop_connect is the identity
        x = q(x)
        if n in FEATURE_BLOCKS:
            feats.append(x)
    return feats


def _basic_conv(sd, name, x, q, padding=0, dilation=1):
    return q(F.selu(_bnf(sd, f'{name}.bn', F.conv2d(x, sd[f'{name}.conv.weight'], padding=padding, dilation=dilation), BN_EPS_DEC)))


def rfb(sd, name, x, q):
    x0 = _basic_conv(sd, f'{name}.branch0.0', x, q)
    outs = [x0]
    for br, kk in ((1, 3), (2, 5), (3, 7)):
        y = _basic_conv(sd, f'{name}.branch{br}.0', x, q)
        y = _basic_conv(sd, f'{name}.branch{br}.1', y, q, padding=(0, kk // 2))
        y = _basic_conv(sd, f'{name}.branch{br}.2', y, q, padding=(kk // 2, 0))
        y = _basic_conv(sd, f'{name}.branch{br}.3', y, q, padding=kk, dilation=kk)
        outs.append(y)
    cat = _basic_conv(sd, f'{name}.conv_cat', torch.cat(outs, 1), q, padding=1)
    return q(torch.relu(cat + _basic_conv(sd, f'{name}.conv_res', x, q)))


def uam(sd, x, q, confidence_ratio=0.1):
    """UnionAttentionModule.forward (att_modules.py:170-187), eval mode (Dropout3d is the identity)"""
    p = 'agg.UAM'
    avg = x.mean((2, 3), keepdim=True)
    xn = _bnf(sd, f'{p}.norm.0', avg, BN_EPS_DEC)
    qc = F.conv2d(xn, sd[f'{p}.channel_q.weight']).squeeze(-1)        # [B, C, 1]
    kc = F.conv2d(xn, sd[f'{p}.channel_k.weight']).squeeze(-1)
    vc = F.conv2d(xn, sd[f'{p}.channel_v.weight']).squeeze(-1)
    att = (torch.softmax(qc @ kc.transpose(1, 2), -1) @ vc).unsqueeze(-1)          # SDPA with scale = 1
    att = torch.sigmoid(F.conv2d(att, sd[f'{p}.fc.weight']))
    xc = q(_bnf(sd, f'{p}.bn', x * att + x, BN_EPS_DEC))
    mask = att.squeeze(3).squeeze(2).clone()
    thr = torch.quantile(mask.float(), confidence_ratio, dim=-1, keepdim=True)
    mask[mask <= thr] = 0.0
    xd = q(xc * mask[:, :, None, None])
    qs = F.conv2d(xd, sd[f'{p}.spatial_q.weight']).squeeze(1)          # [B, H, W]
    ks = F.conv2d(xd, sd[f'{p}.spatial_k.weight']).squeeze(1)
    vs = F.conv2d(xd, sd[f'{p}.spatial_v.weight']).squeeze(1)
    out = (torch.softmax(qs @ ks.transpose(1, 2), -1) @ vs).unsqueeze(1) + vs.unsqueeze(1)
    return q(out)


def aggregation(sd, e4, e3, e2, q):
    up = lambda t: F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=True)
    bc = lambda n, t: _basic_conv(sd, f'agg.{n}', t, q, padding=1)
    e3_1 = q(bc('conv_upsample1', up(e4)) * e3)
    e2_1 = q(bc('conv_upsample2', up(up(e4))) * bc('conv_upsample3', up(e3)) * e2)
    e3_2 = bc('conv_concat2', torch.cat((e3_1, bc('conv_upsample4', up(e4))), 1))
    e2_2 = torch.cat((e2_1, bc('conv_upsample5', up(e3_2))), 1)
    return uam(sd, bc('conv_concat3', e2_2), q)


def _dw(sd, name, x, q, k, padding, dilation):
    y = F.conv2d(x, sd[f'{name}.DWConv.weight'], padding=padding, dilation=dilation, groups=x.shape[1])
    return q(F.selu(_bnf(sd, f'{name}.bn', y, BN_EPS_DEC)))


def object_attention(sd, name, decoder_map, encoder_map, q):
    mask_ob = torch.sigmoid(decoder_map)
    mask_bg = 1 - mask_ob
    x = mask_ob * encoder_map
    edge = mask_bg.clone()
    edge[edge > 0.93] = 0
    x = q(x + edge * encoder_map)
    # DWSConv: depthwise 3x3 + BN + SELU, pointwise + BN + SELU
    y = F.conv2d(x, sd[f'{name}.DWSConv.DWConv.weight'], padding=1, groups=x.shape[1])
    y = q(F.selu(_bnf(sd, f'{name}.DWSConv.bn', y, BN_EPS_DEC)))
    y = q(F.selu(_bnf(sd, f'{name}.DWSConv.bn2', F.conv2d(y, sd[f'{name}.DWSConv.PWConv.weight']), BN_EPS_DEC)))
    skip = y
    parts = []
    for i, (k, pad, dil) in ((1, (1, 0, 1)), (2, (3, 1, 1)), (3, (3, 3, 3)), (4, (3, 5, 5))):
        z = _dw(sd, f'{name}.DWConv{i}.0', y, q, k, pad, dil)
        parts.append(_basic_conv(sd, f'{name}.DWConv{i}.1', z, q))
    y = q(torch.cat(parts, 1) + skip)
    y = torch.relu(_basic_conv(sd, f'{name}.conv1', y, q))
    return q(y + decoder_map)


def decoder(sd, feats, q):
    """TracerDecoder.forward after the encoder (tracer.py:83-97)"""
    x3, x4, x5 = rfb(sd, 'rfb2', feats[1], q), rfb(sd, 'rfb3', feats[2], q), rfb(sd, 'rfb4', feats[3], q)
    d0 = aggregation(sd, x5, x4, x3, q)
    m0 = F.interpolate(d0, scale_factor=8, mode='bilinear')
    d1 = object_attention(sd, 'ObjectAttention2', d0, feats[1], q)
    m1 = F.interpolate(d1, scale_factor=8, mode='bilinear')
    d2 = object_attention(sd, 'ObjectAttention1', q(F.interpolate(d1, scale_factor=2, mode='bilinear')), feats[0], q)
    m2 = F.interpolate(d2, scale_factor=4, mode='bilinear')
    return torch.sigmoid((m2 + m1 + m0) / 3)


def model(sd, x, q=None):
    q = q or (lambda t: t)
    return decoder(sd, encoder(sd, q(x), q), q)


def forward(sd, data, input_image_size=640, erosion=1, batch_size=8, q=None, failure_rule=True):
    """TracerUniversalB7.forward (tracer_b7.py:56-73).  data [N, 3, H, W] in [0, 1].  q: optional rounding applied at every layer boundary
    (emulates the 16-bit module the pipelines run)."""
    q = q or (lambda t: t)
    sd = {k: v.float() for k, v in sd.items()}
    size = (input_image_size, input_image_size) if isinstance(input_image_size, int) else tuple(input_image_size[:2])
    ori = data.shape[-2:]
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    masks = []
    for chunk in q(data.float()).split(batch_size, 0):
        img = F.interpolate(chunk, size=size, mode='bilinear', align_corners=False, antialias=False)     # torchvision Resize(antialias=False)
        img = q((img - mean) / std)
        m = q(model(sd, img, q))
        m = -F.max_pool2d(-m, erosion * 2 + 1, stride=1, padding=erosion)
        masks.append(q(F.interpolate(m, size=tuple(ori), mode='bilinear', align_corners=False, antialias=False)))
    masks = torch.cat(masks, 0)
    if not failure_rule:             # (tests: the map the rule's thresholds are applied to)
        return masks
    failure = (masks > 0.2).flatten(1).all(1)
    masks = masks.masked_fill(failure[:, None, None, None] & (masks < 0.8), 0)
    return masks
