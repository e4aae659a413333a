"""Chip-filling launches of the secondary engines against their oracles.

Round 4 shipped an epilogue that was correct on every 2-image test and wrong in ~1e-5 of the elements of launches with >= 1 tile per CU; the UNet
got its guard in round 5 (rows of the timed 64-image batch, a bitwise all-tiles test).  The VAE decoder, the ControlNet and SRVGGNetCompact run the
same kernels through other instantiations (256 / 128-wide tiles, fused upsample, 64-wide tiles) and had only been compared on 8 x 8 ... 32 x 16
inputs.  Here each runs ONE batch whose launches cover the chip several times over (>= 256 tiles per launch at the widest level) and selected items
are compared with fp32 / half-emulating oracle forwards of those items -- plus a bitwise comparison with the same items run alone (a different
launch geometry for the same arithmetic).  The oracle forwards run on the host cores."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item(), ((a - b).abs().max() / b.abs().max()).item()


def _bars(out, ref16, ref32, what, dtype=torch.float16):
    l2_16, mx_16 = _rel(out, ref16)
    l2_32, _ = _rel(out, ref32)
    emu, _ = _rel(ref16, ref32)
    msg = f'{what}: vs emulated half modules l2={l2_16:.2e} max={mx_16:.2e}; vs fp32 l2={l2_32:.2e}; emulated vs fp32 l2={emu:.2e}'
    print(msg)
    assert torch.isfinite(out).all(), what
    assert l2_16 <= 3e-3 and mx_16 <= 6e-3, msg
    assert l2_32 <= 1.05 * emu + 1e-4, msg


def test_vae_decode_of_a_batch_at_256px(lib):
    """SD VAE topology, 8 latents of 32 x 32 -> 256 x 256 images: the 128-channel level runs 8 x 65536 / 256 = 2048 tiles per launch, the 256-channel
    level 512, the fused-upsample phase convs 512-2048."""
    from oracle import vae_oracle as V
    from mvedit_amd.vae import AutoencoderKLEngine
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    cfg, dtype, B = V.SD_VAE, torch.float16, 8
    sd = {k: v.to(dtype).float() for k, v in V.random_params(cfg, 21).items()}
    eng = AutoencoderKLEngine.from_state_dict(sd, cfg, dtype)
    z = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(5)).to(dtype)
    img = eng.decode(z.cuda(), return_dict=False)[0]
    assert img.shape == (B, 3, 256, 256)
    rows = [0, B - 1]
    with torch.no_grad():
        d32, d16 = V.decode(sd, cfg, z[rows].float()), V.decode(sd, cfg, z[rows].float(), V.quantizer(dtype))
    _bars(img[rows], d16, d32, 'vae decode, rows of a chip-filling batch')
    alone = eng.decode(z[rows].cuda(), return_dict=False)[0]
    assert torch.equal(alone, img[rows]), 'items alone == items of the batch, bit for bit'
    # encode of the same batch of images (stride-2 convs on 256-row tiles)
    x = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(6)) * 2 - 1).to(dtype)
    par = eng.encode(x.cuda(), return_dict=False)[0].parameters
    with torch.no_grad():
        m32, m16 = V.encode_moments(sd, cfg, x[rows].float()), V.encode_moments(sd, cfg, x[rows].float(), V.quantizer(dtype))
    _bars(par[rows], m16, m32, 'vae encode, rows of a chip-filling batch')


def test_controlnet_of_a_16_image_batch_at_full_size(lib):
    """SD-1.5 ControlNet, 16 images at 64 x 64 latents (512 x 512 conditioning images): level 0 = 16 x 4096 / 256 = 256 tiles of 320 columns per launch,
    the conditioning embedding's image-resolution convs 16 x 262144 rows."""
    from oracle import unet_oracle as U
    from mvedit_amd.controlnet import ControlNetEngine
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    cfg, dtype, B, S = U.SD15, torch.float16, 16, 64
    sd = {k: v.to(dtype).float() for k, v in U.make_controlnet_state_dict(cfg, seed=3).items()}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 4, S, S, generator=g).to(dtype)
    ctx = torch.randn(B, 77, 768, generator=g).to(dtype)
    cond = torch.rand(B, 3, 8 * S, 8 * S, generator=g).to(dtype)
    eng = ControlNetEngine.from_state_dict(sd, cfg, dtype)
    down, mid = eng(x.cuda(), 300, ctx.cuda(), cond.cuda(), conditioning_scale=0.7)
    rows = [0, B - 1]
    with torch.no_grad():
        d32, m32 = U.controlnet_forward(sd, cfg, x[rows].float(), 300, ctx[rows].float(), cond[rows].float(), 0.7)
        d16, m16 = U.controlnet_forward(sd, cfg, x[rows].float(), 300, ctx[rows].float(), cond[rows].float(), 0.7, q=U.quantizer(dtype))
    for i, (got, r16, r32) in enumerate(zip(list(down) + [mid], list(d16) + [m16], list(d32) + [m32])):
        _bars(got[rows], r16, r32, f'controlnet output {i}, rows of a chip-filling batch')


def test_srvgg_of_a_batch_at_128px(lib):
    """SRVGGNetCompact (64 features, x4): 8 images of 128 x 128 -> 512 x 512; every 64 -> 64 conv is a launch of 8 x 16384 / 128 = 1024 blocks."""
    from oracle import srvgg_oracle as S
    from mvedit_amd.image_enhancer import SRVGGNetCompactEngine
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    kw, dtype, B = dict(num_feat=64, num_conv=16, upscale=4), torch.float16, 8
    sd = {k: v.to(dtype).float() for k, v in S.random_params(seed=5, **kw).items()}
    x = torch.rand(B, 3, 128, 128, generator=torch.Generator().manual_seed(2)).to(dtype)
    eng = SRVGGNetCompactEngine(3, 3, dtype=dtype, **kw).load_state_dict(sd)
    out = eng(x.cuda())
    assert out.shape == (B, 3, 512, 512)
    rows = [0, B - 1]
    with torch.no_grad():
        y32, y16 = S.forward(sd, x[rows].float(), 4), S.forward(sd, x[rows].float(), 4, q=lambda t: t.to(dtype).float())
    _bars(out[rows], y16, y32, 'srvgg, rows of a chip-filling batch')
    assert torch.equal(eng(x[rows].cuda()), out[rows]), 'items alone == items of the batch, bit for bit'
