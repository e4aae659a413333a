"""SRVGGNetCompact (`image_enhancer` of the pipelines, lib/models/decoders/image_space_ss.py:8-70) on the
native executor vs an oracle that is pinned by the output of the REFERENCE class itself (tests/golden/srvgg_ref.npz, written by
executing the reference's class definition: tests/golden/make_srvgg_golden.py).
Parity bar as tests/test_unet.py: at least as close to fp32 as the emulated half-precision module, <= 3e-3 vs the emulation."""
import os

import numpy as np
import pytest
import torch

from oracle import srvgg_oracle as S

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'srvgg_ref.npz')
CASES = dict(small=dict(num_feat=64, num_conv=3, upscale=4), odd=dict(num_feat=32, num_conv=2, upscale=2))


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item(), ((a - b).abs().max() / b.abs().max()).item()


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize('tag', ['small', 'odd'])
def test_oracle_matches_reference_output(tag):
    g = np.load(GOLDEN)
    kw = CASES[tag]
    with torch.no_grad():
        y = S.forward(S.random_params(seed=7, **kw), torch.from_numpy(g[f'{tag}_x']), kw['upscale'])
    np.testing.assert_allclose(y.numpy(), g[f'{tag}_y'], rtol=1e-5, atol=1e-6)


def test_plan_flops_and_inventory(lib):
    """Plan-time only: the production net (64 features, 32 body convs, x4) costs 2 * 9 * (3*64 + 32*64*64 + 64*48) MAC-FLOPs per input
    pixel; the parameter inventory equals the module's state dict."""
    import ctypes
    from mvedit_amd import _lib
    from mvedit_amd.image_enhancer import SRVGGNetCompactEngine
    eng = SRVGGNetCompactEngine(3, 3, 64, 32, 4, dtype=torch.float16, device='cpu')
    info = eng.plan(6, 128, 128)
    want = 6 * 128 * 128 * 2 * 9 * (8 * 64 + 32 * 64 * 64 + 64 * 48)          # conv_in counts its 8 padded input channels
    assert abs(info['flops']['conv'] - want) < 1e-6 * want and info['n_ops'] == 1 + 33 * 2 + 1 + 1
    buf = ctypes.create_string_buffer(256)
    assert _lib.raw('mve_unet_missing_params')(eng._h, buf, 256) == len(S.param_shapes(3, 3, 64, 32, 4))
    from mvedit_amd import synthetic as SY
    assert SY.srvgg_param_shapes(3, 3, 64, 32, 4) == S.param_shapes(3, 3, 64, 32, 4)
    with pytest.raises(_lib.MveError):
        SRVGGNetCompactEngine(3, 1, 64, 2, 4, dtype=torch.float16, device='cpu')          # the residual needs out == in channels


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('tag,dtype', [('small', torch.float16), ('odd', torch.float16), ('small', torch.bfloat16)])
def test_engine_vs_reference_output(lib, tag, dtype):
    from mvedit_amd.image_enhancer import SRVGGNetCompactEngine
    g = np.load(GOLDEN)
    kw = CASES[tag]
    sd = {k: v.to(dtype).float() for k, v in S.random_params(seed=7, **kw).items()}
    x = torch.from_numpy(g[f'{tag}_x']).to(dtype).float()
    with torch.no_grad():
        y32, y16 = S.forward(sd, x, kw['upscale']), S.forward(sd, x, kw['upscale'], q=lambda t: t.to(dtype).float())
    eng = SRVGGNetCompactEngine(3, 3, dtype=dtype, **kw).load_state_dict(sd)
    out = eng(x.to(dtype).cuda())
    assert out.dtype == dtype and out.shape == y32.shape and torch.isfinite(out).all()
    l2_16, mx_16 = _rel(out, y16)
    l2_32, _ = _rel(out, y32)
    emu, _ = _rel(y16, y32)
    msg = f'vs emulated half module: l2={l2_16:.2e} max={mx_16:.2e}; vs fp32: l2={l2_32:.2e}; emulated vs fp32: l2={emu:.2e}'
    print(msg)
    tol = 3e-3 if dtype == torch.float16 else 2.4e-2
    assert l2_16 <= tol and mx_16 <= 2 * tol, msg
    assert l2_32 <= 1.05 * emu + 1e-4, msg
    # and directly against what the reference class produced from the unrounded weights / input (fp32 module)
    assert _rel(out, torch.from_numpy(g[f'{tag}_y']))[0] < (4e-3 if dtype == torch.float16 else 3e-2)


@pytest.mark.gpu
def test_fp32_io_and_batch_invariance(lib):
    from mvedit_amd.image_enhancer import SRVGGNetCompactEngine
    kw = CASES['small']
    sd = S.random_params(seed=9, **kw)
    eng = SRVGGNetCompactEngine(3, 3, dtype=torch.float16, **kw).load_state_dict({'params': sd})
    x = torch.rand(3, 3, 16, 16, generator=torch.Generator().manual_seed(1)).cuda()
    y = eng(x)
    assert y.dtype == torch.float32 and y.shape == (3, 3, 64, 64)
    assert torch.equal(y[1:2], eng(x[1:2]))
    with torch.no_grad():
        assert _rel(y, S.forward(sd, x.cpu(), 4))[0] < 3e-3
