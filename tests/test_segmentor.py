"""TRACER-B7 foreground segmentor (second half): lib/models/segmentors/tracer_b7.py:16-73 over
lib/models/architecture/tracerb7/.  CPU: the oracle restatement equals the outputs of the REFERENCE modules executed from /root/reference
(tests/golden/tracer_ref.npz, written by tests/golden/make_tracer_golden.py) bit for bit in fp32; the product's synthetic state dict is the
oracle's inventory.  GPU: the HIP engine (mvedit_amd.segmentor, kernels csrc/tracer.hip + mve_gemm + mve_conv3x3) against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import tracer_oracle as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tracer_ref.npz')


def _img(sd, x, size):
    import torch.nn.functional as F
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (F.interpolate(x, size=(size, size), mode='bilinear', align_corners=False) - mean) / std


def test_oracle_equals_reference_executed_golden():
    g = np.load(GOLD)
    sd = T.random_params(seed=11)
    sdf = {k: v.float() for k, v in sd.items()}
    ident = lambda t: t
    for tag, size in (('s192', 192), ('s256', 256)):
        x = torch.from_numpy(g[f'{tag}_x'])
        with torch.no_grad():
            mask = T.forward(sd, x, input_image_size=size, erosion=1, batch_size=2)
            feats = T.encoder(sdf, _img(sd, x[:2], size), ident)
            raw = T.decoder(sdf, feats, ident)
        assert torch.equal(mask, torch.from_numpy(g[f'{tag}_mask'])), tag            # whole wrapper incl. erosion, resize, failure rule
        assert torch.equal(raw, torch.from_numpy(g[f'{tag}_raw']))
        for i, f in enumerate(feats):
            assert torch.equal(f[:, :8], torch.from_numpy(g[f'{tag}_feat{i}']))
            st = g[f'{tag}_feat{i}_stat']
            assert abs(float(f.mean()) - st[0]) < 1e-6 and abs(float(f.std()) - st[1]) < 1e-6
        assert 0.2 < float(raw.max() - raw.min()), 'the seeded network must not collapse to a constant mask'


def test_block_table_and_inventory():
    from mvedit_amd import segmentor as SG, synthetic as S
    assert SG.block_table() == T.block_table()
    stem_pad, blocks = T.block_table()
    assert stem_pad == (0, 1) and len(blocks) == 55
    assert [blocks[i][4] for i in T.FEATURE_BLOCKS] == list(T.FEAT_CH)
    # the stride-2 stages: 300 -> 150 (k3: pad 0/1), 150 -> 75 (k5: 1/2), 75 -> 38 (k3 on an odd size: 1/1), 38 -> 19 (k5: 1/2)
    s2 = [(b[0], b[6]) for b in blocks if b[1] == 2]
    assert s2 == [(3, (0, 1)), (5, (1, 2)), (3, (1, 1)), (5, (1, 2))], s2
    a, b = S.make_tracer_state_dict(3), T.random_params(3)
    assert list(a.keys()) == list(b.keys()) and all(torch.equal(a[k], b[k]) for k in a)
    assert sum(v.numel() for v in a.values()) > 66e6                               # EfficientNet-B7 + decoder: 66 M parameters


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_engine_vs_oracle(lib, dtype):
    """The engine's raw sigmoid map and final masks against the fp32 oracle and against the oracle with 16-bit rounding at every layer
    boundary (what the reference's `.to(dtype)` module computes): the engine rounds less often (BatchNorm folded, fused activations), so it
    must be at least as close to fp32 as the emulated module, and within a small absolute bar of it (masks live in [0, 1])."""
    from mvedit_amd.segmentor import TracerUniversalB7Engine
    from mvedit_amd import synthetic as S
    sd = S.make_tracer_state_dict(11)
    g = np.load(GOLD)
    x = torch.from_numpy(g['s192_x'])
    size = 192
    sdq = {k: v.to(dtype).float() for k, v in sd.items()}
    q = lambda t: t.to(dtype).float()
    with torch.no_grad():
        img = _img(sd, x, size)
        raw32 = T.model({k: v.float() for k, v in sd.items()}, img)
        raw16 = T.model(sdq, img, q)
        mask32 = T.forward(sd, x, input_image_size=size, erosion=1, batch_size=2)
        pre32 = T.forward(sd, x, input_image_size=size, erosion=1, batch_size=2, failure_rule=False)
    eng = TracerUniversalB7Engine(input_image_size=size, batch_size=2, torch_dtype=dtype, erosion=1).load_state_dict(sd)
    raw = eng.raw_mask(x.cuda()).float().cpu()[:, None]
    assert raw.shape == raw32.shape and torch.isfinite(raw).all()
    e_eng, e_emu = float((raw - raw32).abs().max()), float((raw16 - raw32).abs().max())
    m_eng, m_emu = float((raw - raw32).abs().mean()), float((raw16 - raw32).abs().mean())
    print(f'{dtype}: raw map max|engine - fp32| {e_eng:.2e} (emulated 16-bit module {e_emu:.2e}); mean {m_eng:.2e} ({m_emu:.2e})')
    assert m_eng <= 1.5 * m_emu + 2e-3, (m_eng, m_emu)
    assert e_eng <= 3 * e_emu + 2e-2, (e_eng, e_emu)
    masks = eng(x.cuda()).float().cpu()
    assert masks.shape == mask32.shape and masks.dtype == torch.float32
    # masks: the failure rule zeroes pixels below 0.8 -- a pixel within rounding of the threshold may fall on either side; compare away from it
    assert float(pre32.min()) > 0.25, 'every pixel of the seeded network is above 0.2: the failure rule applies to all images (both sides must agree)'
    diff = (masks - mask32).abs()
    near = (pre32 - 0.8).abs() < 0.05
    assert float(diff[~near].max()) <= 3 * e_emu + 3e-2, float(diff[~near].max())
    assert float(near.float().mean()) < 0.5


def _views(n, side, seed):
    """Synthetic 'rendered views': a few soft blobs over a gradient background plus fine noise, in [0, 1] (what do_segmentation hands over)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(n, 3, 10, 10, generator=g)
    x = F.interpolate(low, size=(side, side), mode='bicubic', align_corners=False)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, side), torch.linspace(-1, 1, side), indexing='ij')
    blob = torch.exp(-((xx * 1.4) ** 2 + (yy * 1.1) ** 2) * 3.0)[None, None]
    x = 0.55 * x + 0.35 * blob + 0.1 * torch.rand(n, 3, side, side, generator=g)
    return x.clamp(0, 1)


_FP32_640 = {}


def _oracle_fp32_640(sd, x, size):
    """fp32 oracle outputs at 640^2, computed once for both dtypes (20 s of host time each)."""
    if 'v' not in _FP32_640:
        with torch.no_grad():
            raw32 = T.model({k: v.float() for k, v in sd.items()}, _img(sd, x, size))
            mask32 = T.forward(sd, x, input_image_size=size, erosion=1, batch_size=2)
            pre32 = T.forward(sd, x, input_image_size=size, erosion=1, batch_size=2, failure_rule=False)
        _FP32_640['v'] = (raw32, mask32, pre32)
    return _FP32_640['v']


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_engine_vs_oracle_at_640(lib, dtype):
    """The configuration the pipelines and bench.py run (tracer_b7.py:21 input_image_size=640, bf16, 512^2 views): encoder stages at 160^2 / 80^2 /
    40^2 / 20^2 and the split-K choices that go with them -- none of which the 192^2 case above reaches.  Same statement as there with tighter
    factors and the excluded pixels counted: the raw map is at least as close to the fp32 oracle as the 16-bit-emulated reference module
    (mean within 1.25x, max within 2x), and the final masks (resize -> failure rule at 0.8 -> erosion) agree with the oracle's on every pixel
    that is not within 0.02 of the rule's threshold (at most 10 % of the pixels may be excused; the seeded network has ~0.02 % there)."""
    from mvedit_amd.segmentor import TracerUniversalB7Engine
    from mvedit_amd import synthetic as S
    size = 640
    sd = S.make_tracer_state_dict(11)
    x = _views(2, 512, seed=3)
    q = lambda t: t.to(dtype).float()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    raw32, mask32, pre32 = _oracle_fp32_640(sd, x, size)
    with torch.no_grad():
        raw16 = T.model({k: v.to(dtype).float() for k, v in sd.items()}, _img(sd, x, size), q)
    eng = TracerUniversalB7Engine(input_image_size=size, batch_size=8, torch_dtype=dtype, erosion=1).load_state_dict(sd)
    raw = eng.raw_mask(x.cuda()).float().cpu()[:, None]
    assert raw.shape == raw32.shape == (2, 1, size, size) and torch.isfinite(raw).all()
    e_eng, e_emu = float((raw - raw32).abs().max()), float((raw16 - raw32).abs().max())
    m_eng, m_emu = float((raw - raw32).abs().mean()), float((raw16 - raw32).abs().mean())
    masks = eng(x.cuda()).float().cpu()
    assert masks.shape == mask32.shape == (2, 1, 512, 512)
    near = (pre32 - 0.8).abs() < 0.02
    diff = (masks - mask32).abs()
    frac_near = float(near.float().mean())
    agree = float((diff <= 3 * e_emu + 3e-2).float().mean())
    print(f'640^2 {dtype}: raw map mean|engine - fp32| {m_eng:.2e} (emulated module {m_emu:.2e}); max {e_eng:.2e} ({e_emu:.2e}); '
          f'pixels within 0.02 of the failure threshold {frac_near:.4f}; masks agreeing overall {agree:.4f}; spread of the raw map {float(raw32.max() - raw32.min()):.3f}')
    assert float(raw32.max() - raw32.min()) > 0.2, 'the seeded network must not collapse to a constant map'
    assert m_eng <= 1.25 * m_emu + 5e-4, (m_eng, m_emu)
    assert e_eng <= 2 * e_emu + 1e-2, (e_eng, e_emu)
    assert float(diff[~near].max()) <= 3 * e_emu + 3e-2, float(diff[~near].max())
    assert frac_near < 0.10, frac_near                       # at most 10 % of the pixels are excused
    assert agree > 0.98, agree


@pytest.mark.gpu
def test_engine_batch_chunks_and_shapes(lib):
    """batch_size chunking (tracer_b7.py:63) gives the same masks as one chunk; non-square callers' sizes are restored."""
    from mvedit_amd.segmentor import TracerUniversalB7Engine
    from mvedit_amd import synthetic as S
    sd = S.make_tracer_state_dict(5)
    x = torch.rand(3, 3, 100, 140, generator=torch.Generator().manual_seed(1))
    a = TracerUniversalB7Engine(input_image_size=128, batch_size=8, torch_dtype='bfloat16').load_state_dict(sd)(x.cuda())                 # one chunk of 3
    b = TracerUniversalB7Engine(input_image_size=128, batch_size=2, torch_dtype='bfloat16', min_chunk=1).load_state_dict(sd)(x.cuda())    # chunks of 2 + 1
    c = TracerUniversalB7Engine(input_image_size=128, batch_size=1, torch_dtype='bfloat16', min_chunk=1).load_state_dict(sd)(x.cuda())    # one view at a time
    assert a.shape == (3, 1, 100, 140) and a.dtype == torch.bfloat16
    assert torch.equal(a, b) and torch.equal(a, c)           # the engine's larger internal chunks (min_chunk = 32) rest on this


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('B,H,W,Cin,N,k,pad,dil,gate,res,act', [
    (2, 40, 40, 64, 32, (1, 1), (0, 0), 1, True, True, 0),       # project: SE gate on the operand, skip, two of four channel fragments
    (3, 23, 17, 48, 288, (1, 1), (0, 0), 1, False, False, 1),    # expand + swish: K tail (48 = 32 + 16), ragged pixel tile
    (2, 20, 20, 2304, 384, (1, 1), (0, 0), 1, True, True, 0),    # deep project: the split-K path (K >= 256, <= 6400 pixels per image)
    (1, 20, 20, 640, 644, (1, 1), (0, 0), 1, False, False, 2),   # N % 64 = 4: the narrow store path
    (2, 20, 20, 128, 128, (1, 7), (0, 3), 1, False, False, 2),   # RFB 1 x 7 (split-K: K = 896)
    (2, 20, 20, 128, 128, (7, 1), (3, 0), 1, False, False, 2),   # RFB 7 x 1
    (2, 40, 40, 64, 64, (3, 3), (5, 5), 5, False, True, 2),      # RFB dilated 3 x 3 with a residual
    (1, 80, 80, 32, 32, (1, 3), (0, 1), 1, False, False, 2),     # RFB 1 x 3 at 80^2: no split (K = 96)
])
def test_mconv_vs_torch(lib, dtype, B, H, W, Cin, N, k, pad, dil, gate, res, act):
    """mve_seg_mconv (matrix-core convolution with taps, SE gate, bias, activation, residual fused) against torch fp32 on the same 16-bit
    operands; and an image's result must not depend on the batch it is launched with (bitwise)."""
    import torch.nn.functional as F
    from mvedit_amd import _lib
    from mvedit_amd.ops import dt as _dt
    g = torch.Generator().manual_seed(B * 1000 + N + Cin + k[0] * 7 + k[1])
    dev = torch.device('cuda:0')
    x = torch.randn(B, H, W, Cin, generator=g).to(dtype)
    w = (torch.randn(N, k[0], k[1], Cin, generator=g) * (k[0] * k[1] * Cin) ** -0.5).to(dtype)
    bias = torch.randn(N, generator=g) * 0.3
    gt = torch.rand(B, Cin, generator=g) if gate else None
    rs = torch.randn(B, H, W, N, generator=g).to(dtype) if res else None
    xs = x.float() * gt[:, None, None, :] if gate else x.float()
    xs = xs.to(dtype).float()                                  # the gated operand is rounded to the storage type, as x * gate is in the reference
    ref = F.conv2d(xs.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), bias.double(), padding=pad, dilation=dil).permute(0, 2, 3, 1)
    ref = {0: lambda t: t, 1: F.silu, 2: F.selu}[act](ref)
    if res:
        ref = ref + rs.double()

    def run(xb, gb, rb, nb):
        out = torch.empty(nb * H * W, N + 8, dtype=dtype, device=dev)[:, 4:4 + N] if N % 8 else torch.empty(nb * H * W, N, dtype=dtype, device=dev)
        ldo = out.stride(0)
        with torch.cuda.device(dev):
            _lib.call('mve_seg_mconv', _dt(dtype), _lib.ptr(xb), nb, H, W, Cin, Cin, _lib.ptr(wd), k[0] * k[1] * Cin, k[0], k[1], dil, pad[0], pad[1],
                      _lib.ptr(bd), _lib.ptr(gb), _lib.ptr(rb), N, _lib.ptr(out), N, ldo, act, _lib.stream_ptr(dev))
        return out

    wd, bd = w.reshape(N, -1).contiguous().to(dev), bias.to(dev)
    xd = x.reshape(B * H * W, Cin).to(dev)
    gd = gt.to(dev) if gate else None
    rd = rs.reshape(B * H * W, N).to(dev) if res else None
    out = run(xd, gd, rd, B)
    torch.cuda.synchronize()
    err = (out.double().cpu() - ref.reshape(B * H * W, N)).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= (1.2e-2 if dtype == torch.bfloat16 else 1.6e-3) * max(scale, 1.0), (err, scale)
    b0 = B - 1                                                   # the last image alone
    sl = slice(b0 * H * W, B * H * W)
    one = run(xd[sl].contiguous(), gd[b0:].contiguous() if gate else None, rd[sl].contiguous() if res else None, 1)
    assert torch.equal(one, out[sl])


def test_do_segmentation_host_logic_equals_reference_function():
    """mvedit_amd.pipelines.utils.do_segmentation / Adapter3DMixin.get_tgt_masks against the reference's own functions EXECUTED
    (lib/pipelines/utils.py:73-107, adapter3d_mixin.py:14-19, cut out with ast) over a stand-in segmentor; skipped where /root/reference is absent."""
    import ast
    import types
    import torch.nn.functional as F
    ref = '/root/reference/lib/pipelines/utils.py'
    if not os.path.exists(ref):
        pytest.skip('reference tree not present (GPU box)')
    ns = dict(np=np, torch=torch, F=F)
    for node in ast.parse(open(ref).read()).body:
        if isinstance(node, ast.FunctionDef) and node.name == 'do_segmentation':
            exec(compile(ast.Module([node], []), ref, 'exec'), ns)
    from mvedit_amd.pipelines.utils import do_segmentation
    from mvedit_amd.pipelines.adapter3d_mixin import Adapter3DMixin

    class Seg(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([0.6, -0.2, 0.4]))

        def forward(self, x):
            return torch.sigmoid((x * self.w.view(1, 3, 1, 1)).sum(1, keepdim=True) * 3 - 1)
    seg = Seg()
    x = torch.rand(3, 3, 20, 24, generator=torch.Generator().manual_seed(0))
    x[0, :, :8] = 1.0                                    # a white (background-coloured) band
    for pad, bg in ((0, None), (4, None), (3, (1.0, 1.0, 1.0))):
        want = ns['do_segmentation'](x, seg, padding=pad, bg_color=bg)
        got = do_segmentation(x, seg, padding=pad, bg_color=bg)
        assert torch.equal(got, want), (pad, bg)
    host = types.SimpleNamespace(segmentation=seg, bg_color=(1.0, 1.0, 1.0))
    imgs = x.permute(0, 2, 3, 1)[None] * 1.2 - 0.1
    got = Adapter3DMixin.get_tgt_masks(host, imgs, 3)
    want = ns['do_segmentation'](imgs.squeeze(0).clip(min=0, max=1).permute(0, 3, 1, 2), seg, padding=3, bg_color=(1.0, 1.0, 1.0))[:, 3][None, ..., None]
    assert torch.equal(got, want) and got.shape == (1, 3, 20, 24, 1)
