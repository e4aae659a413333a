"""TRACER-B7 foreground segmentor on the HIP kernels: host mirror of the reference's `TracerUniversalB7`
(lib/models/segmentors/tracer_b7.py:16-73; EfficientNet-B7 encoder lib/models/architecture/tracerb7/efficientnet.py, decoder tracer.py /
att_modules.py / conv_modules.py), the mask model the 3D pipelines run on every denoised view (lib/pipelines/adapter3d_mixin.py:14-19).

Same constructor arguments and call convention as the reference class: `TracerUniversalB7Engine(input_image_size, batch_size, torch_dtype,
erosion)`, `load_state_dict(sd)` with the reference module's own parameter names (`model.` prefix optional), `engine(data)` with
data [N, 3, H, W] in [0, 1] -> masks [N, 1, H, W].  PyTorch tensors are storage only: every operator is a kernel behind the C ABI --
1x1 convolutions are `mve_gemm` on NHWC rows, the decoder's dense 3x3 convolutions `mve_conv3x3`, everything else `mve_seg_*`
(csrc/tracer.hip).  BatchNorm (eval mode) is folded into the neighbouring convolution when the state dict is loaded.  There is no CPU path.
"""
import math

import torch

from . import _lib
from .ops import dt as _dt

ACT_NONE, ACT_SWISH, ACT_SELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3, 4
SELU_SCALE = 1.0507009873554805
BN_EPS_ENC, BN_EPS_DEC = 1e-3, 1e-5
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
    o = math.ceil(size / s)
    p = max((o - 1) * s + (k - 1) + 1 - size, 0)
    return (p // 2, p - p // 2), o


def block_table(image_size=600):
    """(stem padding, [(kernel, stride, expand, cin, cout, se_channels, (pad_before, pad_after))]) of EfficientNet-B7's 55 MBConv blocks.
    width 2.0 / depth 3.1 (efficientnet.py:272-279); the TensorFlow-"SAME" paddings are the static ones computed for a 600-pixel image at
    construction (effi_utils.py:270-315), whatever size is fed later."""
    base = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
    stem_pad, size = _same_pad(image_size, 3, 2)
    blocks = []
    for (r, k, s, e, i, o) in base:
        i, o = _round_filters(i), _round_filters(o)
        for j in range(int(math.ceil(3.1 * r))):
            cin, st = (i, s) if j == 0 else (o, 1)
            pad, size2 = _same_pad(size, k, st)
            blocks.append((k, st, e, cin, o, max(1, int(cin * 0.25)), pad))
            size = size2
    return stem_pad, blocks


class PadHW:
    """explicit (top, bottom, left, right) padding of an asymmetric kernel"""

    def __init__(self, t, b, l, r):
        self.t, self.b, self.l, self.r = t, b, l, r


class TracerUniversalB7Engine:
    def __init__(self, input_image_size=640, batch_size=8, torch_dtype='bfloat16', erosion=1, device='cuda', min_chunk=32, pretrained=None,
                 freeze=True):
        # pretrained / freeze: the reference's constructor arguments (tracer_b7.py:16-30; a config built for the reference constructs this class
        # too).  `pretrained` is a state-dict path the caller may load and hand to load_state_dict (no download here); the engine is inference-only,
        # i.e. always frozen.
        self.pretrained, self.freeze = pretrained, freeze
        self.input_image_size = tuple(input_image_size[:2]) if isinstance(input_image_size, (list, tuple)) else (input_image_size, input_image_size)
        assert self.input_image_size[0] % 32 == 0 and self.input_image_size[1] % 32 == 0, 'input_image_size must be a multiple of 32'
        self.batch_size = batch_size
        # the reference's batch_size is a memory knob of a 24 GB card; an image's result does not depend on its chunk (bitwise, tested), so by
        # default the engine walks chunks of at least 32 views: the deep 20 x 20 layers are latency-bound at 8 views and 4 x cheaper per view
        # at 32 -- for up to 4 x the activation memory batch_size asks for.  min_chunk=1 honours the caller's batch_size exactly.
        self.chunk = max(int(batch_size), int(min_chunk))
        self.dtype = getattr(torch, torch_dtype) if isinstance(torch_dtype, str) else torch_dtype
        assert self.dtype in (torch.float16, torch.bfloat16)
        self.erosion = erosion
        self.device = torch.device(device)
        self.stem_pad, self.blocks = block_table()
        self.p = None
        self._mean_vec = torch.tensor([0.485, 0.456, 0.406], device=self.device)         # transforms.Normalize (tracer_b7.py:42)
        self._std_vec = torch.tensor([0.229, 0.224, 0.225], device=self.device)

    # ------------------------------------------------------------------------------------------------------------------ parameters
    def load_state_dict(self, sd):
        sd = {(k[6:] if k.startswith('model.') else k): v.detach().float().cpu() for k, v in sd.items() if not k.endswith('num_batches_tracked')}
        dev, dt16 = self.device, self.dtype
        P = {}

        def fold(w, bn, eps):
            """conv weight [O, I, kh, kw] + BatchNorm -> (weight, bias) with the affine folded in (fp32)"""
            s = sd[f'{bn}.weight'] / torch.sqrt(sd[f'{bn}.running_var'] + eps)
            return w * s.view(-1, 1, 1, 1), sd[f'{bn}.bias'] - sd[f'{bn}.running_mean'] * s

        f32 = lambda t: t.contiguous().to(dev, torch.float32)
        dense = lambda w: f32(w.permute(0, 2, 3, 1))                              # [O, I, kh, kw] -> [O][kh][kw][I]
        dwise = lambda w: f32(w[:, 0].permute(1, 2, 0))                            # [C, 1, kh, kw] -> [kh][kw][C]
        mat16 = lambda w: w[:, :, 0, 0].contiguous().to(dev, dt16)                # 1x1 conv -> GEMM weight [N][K]

        w, b = fold(sd['encoder._conv_stem.weight'], 'encoder._bn0', BN_EPS_ENC)
        P['stem'] = (dense(w), f32(b))
        for n, (k, st, e, cin, cout, se, pad) in enumerate(self.blocks):
            pre = f'encoder._blocks.{n}'
            blk = {}
            if e != 1:
                w, b = fold(sd[f'{pre}._expand_conv.weight'], f'{pre}._bn0', BN_EPS_ENC)
                blk['expand'] = (mat16(w), f32(b))
            w, b = fold(sd[f'{pre}._depthwise_conv.weight'], f'{pre}._bn1', BN_EPS_ENC)
            blk['dw'] = (dwise(w), f32(b))
            blk['se'] = (f32(sd[f'{pre}._se_reduce.weight'][:, :, 0, 0]), f32(sd[f'{pre}._se_reduce.bias']),
                         f32(sd[f'{pre}._se_expand.weight'][:, :, 0, 0]), f32(sd[f'{pre}._se_expand.bias']))
            w, b = fold(sd[f'{pre}._project_conv.weight'], f'{pre}._bn2', BN_EPS_ENC)
            blk['project'] = (mat16(w), f32(b))
            P[pre] = blk

        def basic(name, conv3x3_mfma=False, mconv=False):
            w, b = fold(sd[f'{name}.conv.weight'], f'{name}.bn', BN_EPS_DEC)
            if conv3x3_mfma:
                return (w.permute(0, 2, 3, 1).contiguous().to(dev, dt16), f32(b))            # mve_conv3x3: [O][3][3][I], 16-bit
            if mconv:
                return (w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(dev, dt16), f32(b))   # mve_seg_mconv: [O][kh * kw * I], 16-bit
            return (dense(w), f32(b))

        for name in ('rfb2', 'rfb3', 'rfb4'):
            P[f'{name}.branch0.0'] = basic(f'{name}.branch0.0', mconv=True)
            for br in (1, 2, 3):
                for i in range(4):
                    P[f'{name}.branch{br}.{i}'] = basic(f'{name}.branch{br}.{i}', mconv=True)
            P[f'{name}.conv_cat'] = basic(f'{name}.conv_cat', True)
            P[f'{name}.conv_res'] = basic(f'{name}.conv_res', mconv=True)
        for n in ('conv_upsample1', 'conv_upsample2', 'conv_upsample3', 'conv_upsample4', 'conv_upsample5', 'conv_concat2', 'conv_concat3'):
            P[f'agg.{n}'] = basic(f'agg.{n}', True)
        u = 'agg.UAM'
        ns = sd[f'{u}.norm.0.weight'] / torch.sqrt(sd[f'{u}.norm.0.running_var'] + BN_EPS_DEC)
        bs = sd[f'{u}.bn.weight'] / torch.sqrt(sd[f'{u}.bn.running_var'] + BN_EPS_DEC)
        P[u] = dict(ns=f32(ns), nb=f32(sd[f'{u}.norm.0.bias'] - sd[f'{u}.norm.0.running_mean'] * ns),
                    bs=f32(bs), bt=f32(sd[f'{u}.bn.bias'] - sd[f'{u}.bn.running_mean'] * bs),
                    wq=f32(sd[f'{u}.channel_q.weight'][:, :, 0, 0]), wk=f32(sd[f'{u}.channel_k.weight'][:, :, 0, 0]),
                    wv=f32(sd[f'{u}.channel_v.weight'][:, :, 0, 0]), wfc=f32(sd[f'{u}.fc.weight'][:, :, 0, 0]),
                    wqkv=f32(torch.cat([sd[f'{u}.spatial_{c}.weight'] for c in 'qkv'], 0).permute(0, 2, 3, 1)))
        for name in ('ObjectAttention2', 'ObjectAttention1'):
            o = {}
            w, b = fold(sd[f'{name}.DWSConv.DWConv.weight'], f'{name}.DWSConv.bn', BN_EPS_DEC)
            o['dws_dw'] = (dwise(w), f32(b))
            w, b = fold(sd[f'{name}.DWSConv.PWConv.weight'], f'{name}.DWSConv.bn2', BN_EPS_DEC)
            o['dws_pw'] = (dense(w), f32(b))
            for i in (1, 2, 3, 4):
                w, b = fold(sd[f'{name}.DWConv{i}.0.DWConv.weight'], f'{name}.DWConv{i}.0.bn', BN_EPS_DEC)
                o[f'dw{i}'] = (dwise(w), f32(b))
                o[f'pw{i}'] = basic(f'{name}.DWConv{i}.1')
            # relu(selu(v)) = selu_scale * relu(v): the SELU behind conv1 is folded into its weights, the kernel applies the ReLU
            w, b = fold(sd[f'{name}.conv1.conv.weight'], f'{name}.conv1.bn', BN_EPS_DEC)
            o['conv1'] = (dense(w * SELU_SCALE), f32(b * SELU_SCALE))
            P[name] = o
        self.p = P
        return self

    # ------------------------------------------------------------------------------------------------------------------ operator wrappers
    def _conv(self, x, B, H, W, Cin, wb, Cout, k=(1, 1), stride=1, pad=(0, 0), dil=1, depthwise=False, act=ACT_NONE, out=None, ldo=None,
              mul=None, add=None, ld2=0, out_f32=False, ldx=None):
        """mve_seg_conv2d on an NHWC tensor (x may be a channel slice: pass ldx); -> (out, Ho, Wo)"""
        kh, kw = k
        if not isinstance(pad, PadHW):                 # (before, after), the same on both axes
            pad = PadHW(pad[0], pad[1], pad[0], pad[1])
        Ho = (H + pad.t + pad.b - ((kh - 1) * dil + 1)) // stride + 1
        Wo = (W + pad.l + pad.r - ((kw - 1) * dil + 1)) // stride + 1
        pt, pl = pad.t, pad.l
        if out is None:
            out = torch.empty(B * Ho * Wo, Cout, dtype=torch.float32 if out_f32 else self.dtype, device=self.device)
            ldo = Cout
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_conv2d', _dt(self.dtype), _lib.ptr(x), B, H, W, Cin, ldx if ldx is not None else Cin, _lib.ptr(wb[0]), _lib.ptr(wb[1]),
                      _lib.ptr(out), Ho, Wo, Cout, ldo, kh, kw, stride, pt, pl, dil, int(depthwise), act, _lib.ptr(mul), _lib.ptr(add), ld2,
                      int(out_f32), _lib.stream_ptr(self.device))
        return out, Ho, Wo

    def _dwconv_pool(self, x, B, H, W, C, wb, k, stride, pad):
        """depthwise k x k + swish of an MBConv block, with the per-slab channel sums the squeeze needs -> (out, Ho, Wo, sums, nslab)"""
        Ho = (H + pad[0] + pad[1] - k) // stride + 1
        Wo = (W + pad[0] + pad[1] - k) // stride + 1
        out = torch.empty(B * Ho * Wo, C, dtype=self.dtype, device=self.device)
        nslab = _lib.raw('mve_seg_dwconv_slabs')(B, Ho, Wo, C, k, stride)
        sums = torch.empty(B, nslab, C, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_dwconv_pool', _dt(self.dtype), _lib.ptr(x), B, H, W, C, C, _lib.ptr(wb[0]), _lib.ptr(wb[1]), _lib.ptr(out), Ho, Wo, C,
                      k, k, stride, pad[0], pad[0], ACT_SWISH, _lib.ptr(sums), _lib.stream_ptr(self.device))
        return out, Ho, Wo, sums, nslab

    def _pw(self, x, B, HW, wb, act=ACT_NONE, gate=None, residual=None):
        """1 x 1 convolution on the matrix cores with bias, activation, the squeeze-and-excite gate on its input and the skip fused"""
        return self._mconv(x, B, HW, 1, wb, act=act, gate=gate, residual=residual)

    def _mconv(self, x, B, H, W, wb, k=(1, 1), pad=(0, 0), dil=1, act=ACT_NONE, gate=None, residual=None, out=None, ldo=None, ldx=None):
        """mve_seg_mconv: stride-1 'same' convolution, weight [N][kh * kw * Cin] 16-bit -> out [B*H*W, N] (or a channel slice of `out`)"""
        w, bias = wb
        N = w.shape[0]
        Cin = w.shape[1] // (k[0] * k[1])
        if out is None:
            out = torch.empty(B * H * W, N, dtype=self.dtype, device=self.device)
            ldo = N
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_mconv', _dt(self.dtype), _lib.ptr(x), B, H, W, Cin, ldx if ldx is not None else Cin, _lib.ptr(w), w.shape[1], k[0], k[1], dil,
                      pad[0], pad[1], _lib.ptr(bias), _lib.ptr(gate), _lib.ptr(residual), N, _lib.ptr(out), N, ldo, act, _lib.stream_ptr(self.device))
        return out

    def _act(self, x, act):
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_act', _dt(self.dtype), _lib.ptr(x), x.numel(), act, _lib.stream_ptr(self.device))
        return x

    def _gemm(self, a, wb, residual=None):
        from . import ops
        return ops.gemm(a, wb[0], bias=wb[1], residual=residual)

    def _conv3x3(self, x1, B, H, W, wb, x2=None):
        from . import ops
        return ops.conv3x3(x1, wb[0], B, H, W, x2=x2, bias=wb[1], splitk=False)[0]

    def _mean(self, x, B, HW, C):
        out = torch.empty(B, C, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_channel_mean', _dt(self.dtype), _lib.ptr(x), B, HW, C, _lib.ptr(out), _lib.stream_ptr(self.device))
        return out

    def _resize(self, x, B, H, W, C, Ho, Wo, align, in_mode=0, out_f32=False, mean=None, std=None):
        out = torch.empty(B * Ho * Wo, C, dtype=torch.float32 if out_f32 else self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_resize', _dt(self.dtype), _lib.ptr(x), B, H, W, C, _lib.ptr(out), Ho, Wo, int(align), in_mode, int(out_f32),
                      _lib.ptr(mean), _lib.ptr(std), _lib.stream_ptr(self.device))
        return out

    def _mul(self, x, y, z=None):
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_mul', _dt(self.dtype), _lib.ptr(x), _lib.ptr(y), _lib.ptr(z), _lib.ptr(out), x.numel(), _lib.stream_ptr(self.device))
        return out

    # ------------------------------------------------------------------------------------------------------------------ network
    def _encoder(self, x, B, H, W):
        """x: NHWC [B*H*W, 3] normalised image -> the four feature maps [(tensor [B*h*w, C], h, w, C)]"""
        P = self.p
        sp = self.stem_pad
        x, H, W = self._conv(x, B, H, W, 3, P['stem'], 64, k=(3, 3), stride=2, pad=sp, act=ACT_SWISH)
        C = 64
        feats = []
        s = _lib.stream_ptr
        for n, (k, st, e, cin, cout, se, pad) in enumerate(self.blocks):
            blk = P[f'encoder._blocks.{n}']
            inp = x
            mid = cin * e
            if e != 1:
                x = self._pw(x, B, H * W, blk['expand'], act=ACT_SWISH)
            x, H2, W2, sums, nslab = self._dwconv_pool(x, B, H, W, mid, blk['dw'], k, st, pad)
            gate = torch.empty(B, mid, dtype=torch.float32, device=self.device)
            hidden = torch.empty(B * (se + mid), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                w1, b1, w2, b2 = blk['se']
                _lib.call('mve_seg_se_gate', _lib.ptr(sums), nslab, 1.0 / (H2 * W2), B, mid, se, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2),
                          _lib.ptr(hidden), _lib.ptr(gate), s(self.device))
            x = self._pw(x, B, H2 * W2, blk['project'], gate=gate, residual=inp if (st == 1 and cin == cout) else None)
            H, W, C = H2, W2, cout
            if n in FEATURE_BLOCKS:
                feats.append((x, H, W, C))
        return feats

    def _rfb(self, name, feat, B):
        """RFB_Block (att_modules.py:23-72): four branches of 1 x 1 / 1 x k / k x 1 / dilated 3 x 3 BasicConv2d layers, all on the matrix cores"""
        x, H, W, Cin = feat
        P = self.p
        c = P[f'{name}.branch0.0'][0].shape[0]
        M = B * H * W
        cat = torch.empty(M, 4 * c, dtype=self.dtype, device=self.device)
        self._mconv(x, B, H, W, P[f'{name}.branch0.0'], act=ACT_SELU, out=cat, ldo=4 * c)
        for br, kk in ((1, 3), (2, 5), (3, 7)):
            y = self._mconv(x, B, H, W, P[f'{name}.branch{br}.0'], act=ACT_SELU)
            y = self._mconv(y, B, H, W, P[f'{name}.branch{br}.1'], k=(1, kk), pad=(0, kk // 2), act=ACT_SELU)
            y = self._mconv(y, B, H, W, P[f'{name}.branch{br}.2'], k=(kk, 1), pad=(kk // 2, 0), act=ACT_SELU)
            self._mconv(y, B, H, W, P[f'{name}.branch{br}.3'], k=(3, 3), pad=(kk, kk), dil=kk, act=ACT_SELU, out=cat[:, br * c:], ldo=4 * c)
        cc = self._act(self._conv3x3(cat, B, H, W, P[f'{name}.conv_cat']), ACT_SELU)
        out = self._mconv(x, B, H, W, P[f'{name}.conv_res'], act=ACT_SELU, residual=cc)
        return self._act(out, ACT_RELU), H, W, c

    def _aggregation(self, e4, e3, e2, B):
        P = self.p
        (x4, h4, w4, c4), (x3, h3, w3, c3), (x2, h2, w2, c2) = e4, e3, e2
        up = lambda t, h, w, c: self._resize(t, B, h, w, c, 2 * h, 2 * w, True)
        bc = lambda n, t, h, w, t2=None: self._act(self._conv3x3(t, B, h, w, P[f'agg.{n}'], x2=t2), ACT_SELU)
        u4 = up(x4, h4, w4, c4)                                                        # [h3, w3, c4]
        e3_1 = self._mul(bc('conv_upsample1', u4, h3, w3), x3)
        uu4 = up(u4, h3, w3, c4)                                                       # [h2, w2, c4]
        e2_1 = self._mul(bc('conv_upsample2', uu4, h2, w2), bc('conv_upsample3', up(x3, h3, w3, c3), h2, w2), x2)
        e3_2 = bc('conv_concat2', e3_1, h3, w3, bc('conv_upsample4', u4, h3, w3))
        x = bc('conv_concat3', e2_1, h2, w2, bc('conv_upsample5', up(e3_2, h3, w3, c3 + c4), h2, w2))
        return self._uam(x, B, h2, w2, c2 + c3 + c4)

    def _uam(self, x, B, H, W, C):
        u = self.p['agg.UAM']
        pooled = self._mean(x, B, H * W, C)
        att = torch.empty(B, C, dtype=torch.float32, device=self.device)
        A, S = torch.empty_like(att), torch.empty_like(att)
        xd = torch.empty_like(x)
        out = torch.empty(B, H * W, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            st = _lib.stream_ptr(self.device)
            _lib.call('mve_seg_uam_channel', _lib.ptr(pooled), B, C, _lib.ptr(u['ns']), _lib.ptr(u['nb']), _lib.ptr(u['wq']), _lib.ptr(u['wk']),
                      _lib.ptr(u['wv']), _lib.ptr(u['wfc']), 0.1, _lib.ptr(u['bs']), _lib.ptr(u['bt']), _lib.ptr(att), _lib.ptr(A), _lib.ptr(S), st)
            _lib.call('mve_seg_scale', _dt(self.dtype), _lib.ptr(xd), _lib.ptr(x), B, H * W, C, _lib.ptr(A), _lib.ptr(S), st)
        qkv, _, _ = self._conv(xd, B, H, W, C, (u['wqkv'], None), 3, out_f32=True)
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_uam_spatial', _lib.ptr(qkv), B, H, W, _lib.ptr(out), _lib.stream_ptr(self.device))
        return out

    def _object_attention(self, name, dmap, feat, B):
        """dmap f32 [B, H*W]; feat (x, H, W, C) -> f32 [B, H*W]"""
        o = self.p[name]
        enc, H, W, C = feat
        h2 = C // 2
        x = torch.empty_like(enc)
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_object_mix', _dt(self.dtype), _lib.ptr(dmap), _lib.ptr(enc), _lib.ptr(x), B, H * W, C, _lib.stream_ptr(self.device))
        y, _, _ = self._conv(x, B, H, W, C, o['dws_dw'], C, k=(3, 3), pad=(1, 1), depthwise=True, act=ACT_SELU)
        y, _, _ = self._conv(y, B, H, W, C, o['dws_pw'], h2, act=ACT_SELU)                        # skip = y
        cat = torch.empty(B * H * W, h2, dtype=self.dtype, device=self.device)
        c8 = C // 8
        for i, (k, pad, dil) in ((1, (1, 0, 1)), (2, (3, 1, 1)), (3, (3, 3, 3)), (4, (3, 5, 5))):
            z, _, _ = self._conv(y, B, H, W, h2, o[f'dw{i}'], h2, k=(k, k), pad=(pad, pad), dil=dil, depthwise=True, act=ACT_SELU)
            self._conv(z, B, H, W, h2, o[f'pw{i}'], c8, act=ACT_SELU, out=cat[:, (i - 1) * c8:], ldo=h2, add=y[:, (i - 1) * c8:], ld2=h2)
        out, _, _ = self._conv(cat, B, H, W, h2, o['conv1'], 1, act=ACT_RELU, add=dmap, ld2=1, out_f32=True)
        return out.view(B, H * W)

    def _model(self, img, B, S0, S1):
        feats = self._encoder(img, B, S0, S1)
        x3, x4, x5 = self._rfb('rfb2', feats[1], B), self._rfb('rfb3', feats[2], B), self._rfb('rfb4', feats[3], B)
        d0 = self._aggregation(x5, x4, x3, B)                                                   # f32 [B, (S/8)^2]
        d1 = self._object_attention('ObjectAttention2', d0, feats[1], B)
        h8, w8 = S0 // 8, S1 // 8
        ds = self._resize(d1, B, h8, w8, 1, 2 * h8, 2 * w8, False, in_mode=1, out_f32=True).view(B, -1)
        d2 = self._object_attention('ObjectAttention1', ds, feats[0], B)
        out = torch.empty(B, S0 * S1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.call('mve_seg_fuse', _lib.ptr(d0), _lib.ptr(d1), _lib.ptr(d2), B, S0, S1, _lib.ptr(out), _lib.stream_ptr(self.device))
        return out

    @torch.no_grad()
    def __call__(self, data):
        """data [N, 3, H, W] in [0, 1] (any float dtype) -> masks [N, 1, H, W] in the engine dtype (tracer_b7.py:56-73)"""
        assert self.p is not None, 'load_state_dict first'
        N, _, H0, W0 = data.shape
        S0, S1 = self.input_image_size
        masks = torch.empty(N, 1, H0, W0, dtype=self.dtype, device=self.device)
        for i0 in range(0, N, self.chunk):
            chunk = data[i0:i0 + self.chunk].to(self.device, torch.float32).contiguous()
            B = chunk.shape[0]
            img = self._resize(chunk, B, H0, W0, 3, S0, S1, False, in_mode=2, mean=self._mean_vec, std=self._std_vec)
            m = self._model(img, B, S0, S1)
            nbytes = _lib.raw('mve_seg_post_workspace_bytes')(B, S0, S1, H0, W0)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                _lib.call('mve_seg_post', _dt(self.dtype), _lib.ptr(m), B, S0, S1, self.erosion, _lib.ptr(masks[i0:i0 + B]), H0, W0, 0, _lib.ptr(ws),
                          _lib.stream_ptr(self.device))
        return masks

    forward = __call__

    def raw_mask(self, data):
        """The network's sigmoid output at the input resolution ([N, S, S] fp32) without erosion / resize / failure rule (for tests)."""
        N, _, H0, W0 = data.shape
        S0, S1 = self.input_image_size
        chunk = data.to(self.device, torch.float32).contiguous()
        img = self._resize(chunk, N, H0, W0, 3, S0, S1, False, in_mode=2, mean=self._mean_vec, std=self._std_vec)
        return self._model(img, N, S0, S1).view(N, S0, S1)
