"""CPU: the oracle's phase form of Upsample2D (oracle/unet_oracle.py: upsample_phase_weights / upsample_conv_phases -- what the engine's
mve_upsample_conv_phases is checked against on the GPU) IS nearest-2x upsample + conv 3x3 (padding 1), the reference's computation
(diffusers Upsample2D, reached from lib/models/architecture/diffusers.py:57-164), including the borders and odd, non-square sizes."""
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as UO


@pytest.mark.parametrize('B,C,Cout,H,W', [(2, 8, 4, 5, 6), (1, 3, 7, 1, 1), (1, 16, 16, 8, 8), (3, 4, 2, 2, 9)])
def test_phase_form_is_upsample_then_conv(B, C, Cout, H, W):
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, C, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode='nearest'), w, b, padding=1)
    got = UO.upsample_conv_phases(x.float(), w.float(), b.float())
    assert got.shape == ref.shape
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_phase_weights_are_tap_sums_with_one_rounding():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(4, 8, 3, 3, generator=g).half().float()
    w4 = UO.upsample_phase_weights(w)
    # every 3 x 3 tap is counted once in every phase
    for py in (0, 1):
        for px in (0, 1):
            assert torch.allclose(w4[py][px].sum((2, 3)), w.sum((2, 3)), atol=1e-5)
    assert torch.equal(w4[0][0][:, :, 0, 0], w[:, :, 0, 0]) and torch.equal(w4[1][1][:, :, 1, 1], w[:, :, 2, 2])
    assert torch.equal(w4[0][1][:, :, 1, 1], w[:, :, 1, 2] + w[:, :, 2, 2]) and torch.equal(w4[0][0][:, :, 0, 1], w[:, :, 0, 1] + w[:, :, 0, 2])
    q = UO.quantizer(torch.float16)
    w4q = UO.upsample_phase_weights(w, q)
    assert all(torch.equal(w4q[py][px], q(w4[py][px])) for py in (0, 1) for px in (0, 1))


@pytest.mark.parametrize('B,H,W,C', [(1, 1, 1, 8), (2, 4, 8, 16), (3, 8, 2, 24), (2, 2, 256, 8)])
def test_grouped_output_rows_address_the_upsampled_nhwc_image(B, H, W, C):
    """The addressing contract of mve_upsample_conv_phases (include/mvedit_amd.h, GemmParams::orow_* / ConvGeom::phase_rows in csrc/gemm_shared.h),
    restated in integers: row m of the one-launch form (R = B H W rows per phase, m = ph R + (b H + i) W + j) starts at element
        mm * ldc + ((mm >> log2 W) + (ph >> 1)) * extra + (ph & 1) * ldc / 2,   mm = m - ph R, ldc = 2 C, extra = 2 W C
    and that is pixel (b, 2 i + (ph >> 1), 2 j + (ph & 1)) of the dense [B][2H][2W][C] output -- every output element exactly once."""
    lw = W.bit_length() - 1
    assert 1 << lw == W
    R, ldc, extra = B * H * W, 2 * C, 2 * W * C
    seen = torch.zeros(B * 4 * H * W * C, dtype=torch.int32)
    for m in range(4 * R):
        ph = int(m >= R) + int(m >= 2 * R) + int(m >= 3 * R)
        mm = m - ph * R
        off = mm * ldc + ((mm >> lw) + (ph >> 1)) * extra + (ph & 1) * (ldc >> 1)
        b, r = divmod(mm, H * W)
        i, j = divmod(r, W)
        want = (((b * 2 * H) + 2 * i + (ph >> 1)) * 2 * W + 2 * j + (ph & 1)) * C
        assert off == want, (m, ph, b, i, j)
        seen[off:off + C] += 1
    assert bool((seen == 1).all())
