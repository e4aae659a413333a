"""Deterministic stand-ins for `pipe.unet` / `pipe.controlnet` with the diffusers call signatures the reference uses
(lib/pipelines/adapter3d_mixin.py:101-125).  Every output row is a function of the SAME row of the inputs only (plus, under
cross-image attention, of its pair), so chunked and fused walks agree -- like the real networks.  Shared by
tests/golden/make_mixin_golden.py (which runs the REFERENCE's get_noise_pred over them) and tests/test_pipeline_mixin_ref.py."""
import torch


class StubControlNet:
    nets = [None, None]

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode=False,
                 added_cond_kwargs=None, return_dict=False):
        assert not guess_mode and not return_dict and len(controlnet_cond) == len(conditioning_scale)
        b = sample.shape[0]
        drive = sample.mean(dim=(1, 2, 3)) + 0.1 * encoder_hidden_states.mean(dim=(1, 2))
        for k, (c, s) in enumerate(zip(controlnet_cond, conditioning_scale)):
            if c is not None:
                drive = drive + float(s) * (k + 1) * c.reshape(b, -1).mean(dim=1)
        drive = drive + 1e-3 * float(torch.as_tensor(timestep).float().mean())
        down = [drive.view(b, 1, 1, 1) * torch.ones(b, 3, sample.shape[2], sample.shape[3], dtype=sample.dtype) * (j + 1) for j in range(2)]
        mid = -drive.view(b, 1, 1, 1) * torch.ones(b, 5, sample.shape[2] // 2, sample.shape[3] // 2, dtype=sample.dtype)
        return down, mid


class StubUNet:
    def __call__(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, down_block_additional_residuals=None,
                 mid_block_additional_residual=None, added_cond_kwargs=None, return_dict=False):
        assert not return_dict
        b = sample.shape[0]
        out = 0.7 * sample + 0.05 * encoder_hidden_states.mean(dim=(1, 2)).view(b, 1, 1, 1)
        out = out + 1e-4 * float(torch.as_tensor(timestep).float().mean())
        if down_block_additional_residuals is not None:
            for r in down_block_additional_residuals:
                out = out + 0.01 * r.reshape(b, -1).mean(dim=1).view(b, 1, 1, 1)
            out = out + 0.02 * mid_block_additional_residual.reshape(b, -1).mean(dim=1).view(b, 1, 1, 1)
        n = (cross_attention_kwargs or {}).get('num_cross_attn_imgs', 1)
        if n > 1:       # every image of a group sees the group's mean (what joint attention amounts to for this stand-in)
            g = out.reshape(b // n, n, *out.shape[1:])
            out = (g + 0.3 * g.mean(dim=(1, 2, 3, 4), keepdim=True)).reshape(out.shape)
        return (out,)


def cases():
    """name -> kwargs of get_noise_pred (seeded); V views split in chunks as the pipelines do (uncond half first, then text half)."""
    g = torch.Generator().manual_seed(0)
    out = {}
    V, H = 5, 8

    def split(x, bs=2):
        return tuple(x.split(bs, dim=0))
    lat = torch.randn(V, 4, H, H, generator=g)
    emb = torch.randn(2 * V, 7, 16, generator=g)
    img, dep, ext = torch.rand(V, 3, 8 * H, 8 * H, generator=g), torch.rand(V, 3, 8 * H, 8 * H, generator=g), torch.rand(V, 3, 8 * H, 8 * H, generator=g)
    two = lambda x: torch.cat([x, x], dim=0)
    out['plain'] = dict(latent_batches=split(two(lat)), prompt_embeds_batches=split(emb), ctrl_images_batches=split(two(img)),
                        ctrl_depths_batches=split(two(dep)), t=torch.tensor(499), tile_weight=0.6, depth_weight=0.4, guidance_scale=7.0,
                        extra_control_batches=[split(two(ext))])
    out['no_depth_adapter_scale'] = dict(latent_batches=split(two(lat)), prompt_embeds_batches=split(emb), ctrl_images_batches=split(two(img)),
                                         ctrl_depths_batches=None, t=torch.tensor(250), tile_weight=1.0, depth_weight=0.0, guidance_scale=5.0,
                                         adapter_scale=1.5)
    # use_reference: the text half is [reference | view] stacked along H, the uncond half is the view alone (mvedit_3d_pipeline.py:1226-1236)
    ref = torch.randn(V, 4, H, H, generator=g)
    paired = torch.cat([ref, lat], dim=2)
    out['paired'] = dict(latent_batches=split(lat) + split(paired), prompt_embeds_batches=split(emb[:V]) + split(emb[V:]),
                         ctrl_images_batches=split(img) * 2, ctrl_depths_batches=split(dep) * 2, t=torch.tensor(700), tile_weight=0.5,
                         depth_weight=0.5, guidance_scale=7.0)
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# 2-pass mode (lib/pipelines/adapter3d_mixin.py:137-317): stand-ins for MultiControlNetModel (summing per-net outputs, as diffusers
# does), and for unet_enc / unet_dec (lib/models/architecture/diffusers.py:57-164) including the reference-attention kwargs
# (mode 'w' stores a per-row summary in ref_dict, modes 'r' / 'm' read it).
# ---------------------------------------------------------------------------------------------------------------------------
class StubNet:
    def __init__(self, k):
        self.k = k


class StubMulti:
    def __init__(self, nets):
        self.nets = list(nets)

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode=False,
                 added_cond_kwargs=None, return_dict=False):
        assert not return_dict and len(controlnet_cond) == len(conditioning_scale) == len(self.nets)
        b = sample.shape[0]
        drive = torch.zeros(b, dtype=sample.dtype)
        for net, c, s in zip(self.nets, controlnet_cond, conditioning_scale):
            d = (net.k + 1) * c.reshape(b, -1).mean(dim=1) + 0.3 * sample.mean(dim=(1, 2, 3)) + 0.1 * encoder_hidden_states.mean(dim=(1, 2))
            drive = drive + float(s) * (d + (0.5 if guess_mode else 0.0))
        drive = drive + 1e-3 * float(torch.as_tensor(timestep).float().mean())
        down = tuple(drive.view(b, 1, 1, 1) * torch.ones(b, 3, sample.shape[2], sample.shape[3], dtype=sample.dtype) * (j + 1) for j in range(2))
        mid = -drive.view(b, 1, 1, 1) * torch.ones(b, 5, sample.shape[2] // 2, sample.shape[3] // 2, dtype=sample.dtype)
        return list(down), mid


def _mix(x, cak):
    n = (cak or {}).get('num_cross_attn_imgs', 1)
    if n > 1:
        g = x.reshape(x.shape[0] // n, n, *x.shape[1:])
        x = (g + 0.3 * g.mean(dim=tuple(range(1, g.dim())), keepdim=True)).reshape(x.shape)
    return x


def _ref(x, cak, key):
    """reference attention stand-in: 'w' stores this call's per-row mean, 'r' / 'm' add the stored one"""
    if cak and 'mode' in cak:
        if cak['mode'] == 'w':
            cak['ref_dict'][key] = x.reshape(x.shape[0], -1).mean(dim=1)
        else:
            x = x + 0.2 * cak['ref_dict'][key].view(-1, *[1] * (x.dim() - 1))
    return x


def stub_unet_enc(unet, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, added_cond_kwargs=None):
    b = sample.shape[0]
    emb = torch.stack([sample.mean(dim=(1, 2, 3)), encoder_hidden_states.mean(dim=(1, 2)),
                       torch.full((b,), 1e-3 * float(torch.as_tensor(timestep).float().mean()))], dim=1)
    h = _ref(_mix(0.9 * sample, cross_attention_kwargs), cross_attention_kwargs, 'enc')
    down = (h[:, :3], 0.5 * h[:, :3])
    mid = torch.nn.functional.avg_pool2d(h, 2).repeat(1, 2, 1, 1)[:, :5]
    return emb, down, mid


def stub_unet_dec(unet, emb, down_block_res_samples, sample, encoder_hidden_states, cross_attention_kwargs=None,
                  down_block_additional_residuals=None, mid_block_additional_residual=None):
    b = sample.shape[0]
    down = list(down_block_res_samples)
    if down_block_additional_residuals is not None and mid_block_additional_residual is not None:
        down = [a + r for a, r in zip(down, down_block_additional_residuals)]
        sample = sample + mid_block_additional_residual
    out = torch.nn.functional.interpolate(sample[:, :4], scale_factor=2.0, mode='nearest')
    out = out + 0.1 * emb.sum(dim=1).view(b, 1, 1, 1) + 0.05 * encoder_hidden_states.mean(dim=(1, 2)).view(b, 1, 1, 1)
    for d in down:
        out = out + 0.2 * torch.cat([d, d[:, :1]], dim=1)
    return _ref(_mix(out, cross_attention_kwargs), cross_attention_kwargs, 'dec')


def cases_2pass():
    g = torch.Generator().manual_seed(1)
    V, H = 4, 8
    split = lambda x, bs=2: tuple(x.split(bs, dim=0))
    two = lambda x: torch.cat([x, x], dim=0)
    lat = torch.randn(V, 4, H, H, generator=g)
    emb = torch.randn(2 * V, 7, 16, generator=g)
    img, dep, ext = (torch.rand(V, 3, 8 * H, 8 * H, generator=g) for _ in range(3))
    out = {}
    out['plain'] = dict(
        p1=dict(latent_batches=split(two(lat)), prompt_embeds_batches=split(emb), t=torch.tensor(499), guidance_scale=7.0,
                ctrl_depths_batches=split(two(dep)), depth_weight=0.4, extra_control_batches=[split(two(ext))]),
        p2=dict(latent_batches=split(two(lat)), prompt_embeds_batches=split(emb), t=torch.tensor(499), guidance_scale=7.0,
                ctrl_images_batches=split(two(img)), tile_weight=0.6, ctrl_depths_batches=split(two(dep)), depth_weight=0.4))
    ref = torch.randn(V, 4, H, H, generator=g)
    paired = torch.cat([ref, lat], dim=2)
    out['paired'] = dict(
        p1=dict(latent_batches=split(lat) + split(paired), prompt_embeds_batches=split(emb[:V]) + split(emb[V:]), t=torch.tensor(300),
                guidance_scale=5.0, ctrl_depths_batches=split(dep) * 2, depth_weight=0.7),
        p2=dict(latent_batches=split(lat) + split(paired), prompt_embeds_batches=split(emb[:V]) + split(emb[V:]), t=torch.tensor(300),
                guidance_scale=5.0, ctrl_images_batches=split(img) * 2, tile_weight=0.3, ctrl_depths_batches=split(dep) * 2, depth_weight=0.7,
                adapter_scale=1.2))
    cond = torch.randn(V, 4, H, H, generator=g)
    out['reference_attention_no_depth'] = dict(
        p1=dict(latent_batches=split(two(lat)), prompt_embeds_batches=split(emb), t=torch.tensor(100), guidance_scale=3.0,
                cond_noisy_latent_batches=split(two(cond))),
        p2=dict(latent_batches=split(two(lat)), prompt_embeds_batches=split(emb), t=torch.tensor(100), guidance_scale=3.0,
                ctrl_images_batches=split(two(img)), tile_weight=1.0, ctrl_text_embedding=False))
    return out
