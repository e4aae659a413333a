"""ORACLE (test infrastructure only -- never imported by the product path).

torchvision.transforms.functional.gaussian_blur restated in plain torch (torchvision==0.16 `_get_gaussian_kernel1d/2d`, `gaussian_blur`:
float32 kernel from linspace / exp / normalise, outer product, reflect padding by ksize // 2, depthwise conv2d) and the reference's
`highpass` built on it (lib/pipelines/utils.py:187-188).  PARITY UNPINNED: torchvision is absent from this image; the restatement follows
its published source and is cross-checked against an independent dense construction in tests/test_blur.py."""
import torch
import torch.nn.functional as F


def kernel1d(ksize, sigma, dtype=torch.float32):
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize, dtype=dtype)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(img, ksize, sigma):
    """img [..., C, H, W] (any float dtype; the kernel is built in that dtype like torchvision does)"""
    k = kernel1d(ksize, sigma, img.dtype).to(img.device)
    k2 = k[:, None] @ k[None, :]
    lead = img.shape[:-3]
    x = img.reshape(-1, *img.shape[-3:])
    C = x.shape[1]
    r = ksize // 2
    x = F.pad(x, [r, r, r, r], mode='reflect')
    x = F.conv2d(x, k2.expand(C, 1, ksize, ksize), groups=C)
    return x.reshape(*lead, *x.shape[-3:])


def highpass(x, std=5, offset=0.5):
    return offset + x - gaussian_blur(x, int(round(std)) * 6 + 1, std)
