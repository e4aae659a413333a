"""Denoise-step bookkeeping of the reference pipelines: the noise scales of a (possibly fractional)
timestep and the x0 prediction from the guided noise estimate.

    get_noise_scales   lib/core/diffusion.py:4-21
    predict_x0         lib/pipelines/mvedit_3d_pipeline.py:1253-1255:  (latents_scaled - sqrt(1-abar) * noise_pred) / sqrt(abar)

The scheduler objects themselves (diffusers EulerAncestralDiscreteScheduler etc.) are third-party and stay what the runner loaded."""
import torch

from .. import _lib


def get_noise_scales(alphas_bar, t, num_timesteps, dtype=torch.float32):
    """-> (sqrt(abar_t), sqrt(1 - abar_t)); fractional t interpolates linearly in the variance-exploding sigma."""
    ab = torch.as_tensor(alphas_bar, dtype=torch.float32, device=t.device)
    if t.is_floating_point():
        assert ((t >= 0) & (t <= num_timesteps - 1)).all()
        lo = t.long()
        frac = t - lo
        sig = [torch.sqrt((1 - ab[i]) / ab[i]) for i in (lo, (lo + 1).clamp(max=num_timesteps - 1))]
        sigma = sig[0] * (1 - frac) + sig[1] * frac
        s2 = sigma ** 2
        a, b = torch.sqrt(1 / (1 + s2)), torch.sqrt(s2 / (1 + s2))
    else:
        a, b = torch.sqrt(ab[t]), torch.sqrt(1 - ab[t])
    return a.to(dtype), b.to(dtype)


def predict_x0(latents_scaled, noise_pred, sqrt_alpha_bar_t, sqrt_one_minus_alpha_bar_t):
    """x0 = (latents_scaled - sqrt(1-abar) * noise_pred) / sqrt(abar), in noise_pred's dtype; one elementwise launch."""
    assert latents_scaled.shape == noise_pred.shape and latents_scaled.is_cuda
    x = latents_scaled.float().contiguous()
    e = noise_pred.float().contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.call('mve_x0_prediction', _lib.ptr(x), _lib.ptr(e), float(sqrt_alpha_bar_t), float(sqrt_one_minus_alpha_bar_t), x.numel(),
                  _lib.ptr(out), _lib.stream_ptr(x.device))
    return out.to(noise_pred.dtype)
