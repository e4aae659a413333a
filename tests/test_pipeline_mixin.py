"""`Adapter3DMixin.get_noise_pred` mirror (lib/pipelines/adapter3d_mixin.py:68-135): chunked walk == fused
single pass (batch invariance), reference pairing, CFG, against the oracle."""
import pytest
import torch

from oracle import unet_oracle as U

pytestmark = pytest.mark.gpu


class _Pipe:
    pass


def _make(cfg, dtype, with_cn):
    from mvedit_amd.unet import UNet2DConditionEngine
    from mvedit_amd.pipelines import Adapter3DMixin

    class Pipe(Adapter3DMixin):
        pass
    sd = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=21).items()}
    p = Pipe()
    p.unet = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype)
    p.controlnet = None
    if with_cn:
        from test_unet import residuals

        def controlnet(sample, t, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode, added_cond_kwargs,
                       return_dict):
            # stand-in ControlNet: deterministic residuals scaled by the conditioning weights
            down, mid = residuals(cfg, sample.shape[0], sample.shape[-1], seed=5)
            s = float(conditioning_scale[0]) + float(conditioning_scale[1])
            return [(s * d).to(sample.dtype).cuda() for d in down], (s * mid).to(sample.dtype).cuda()
        p.controlnet = controlnet
    return p, sd


def test_get_noise_pred_fused_equals_chunked_and_oracle(lib):
    cfg, dtype = U.TINY, torch.float16
    p, sd = _make(cfg, dtype, with_cn=False)
    V, S = 6, 16
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(V, 4, S, S, generator=g).to(dtype)
    emb_u = torch.randn(1, 77, 768, generator=g).to(dtype).expand(V, -1, -1)
    emb_t = torch.randn(V, 77, 768, generator=g).to(dtype)
    lat2 = torch.cat([lat, lat]).cuda()
    emb2 = torch.cat([emb_u, emb_t]).cuda()
    chunks = lambda x, n: list(x.split(n, dim=0))
    p.fuse_chunks = True
    fused = p.get_noise_pred(chunks(lat2, 4), chunks(emb2, 4), [None] * 3, None, 499, 0.0, 0.0, 7.0)
    p.fuse_chunks = False
    walked = p.get_noise_pred(chunks(lat2, 4), chunks(emb2, 4), [None] * 3, None, 499, 0.0, 0.0, 7.0)
    assert fused.shape == (V, 4, S, S)
    assert torch.equal(fused, walked), 'the UNet engine must be batch-invariant'
    with torch.no_grad():
        ref = U.unet_forward(sd, cfg, lat2.float().cpu(), 499, emb2.float().cpu(), q=U.quantizer(dtype))
    halves = ref
    ref = 7.0 * ref[V:] + (1 - 7.0) * ref[:V]
    # CFG is linear: an error eps on each half becomes at most (g + |1-g|) * eps on the combination
    bound = (7.0 + 6.0) * 2e-3 * max(halves[V:].norm(), halves[:V].norm()).item()
    err = (fused.float().cpu() - ref).norm().item()
    assert err <= bound, (err, bound)
    adapter = p.get_noise_pred(chunks(lat2, 4), chunks(emb2, 4), [None] * 3, None, 499, 0.0, 0.0, 7.0, adapter_scale=2.0)
    assert adapter.shape == fused.shape


def test_get_noise_pred_reference_pairing_with_controlnet(lib):
    """use_reference: latents are [b, 4, 2H, W] (reference image on top), cross-image attention, zero ControlNet
    residuals for the reference rows, only the view half is returned (adapter3d_mixin.py:86-127)."""
    cfg, dtype = U.TINY, torch.float16
    p, sd = _make(cfg, dtype, with_cn=True)
    V, S = 2, 16
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(2 * V, 4, 2 * S, S, generator=g).to(dtype).cuda()
    emb = torch.randn(2 * V, 77, 768, generator=g).to(dtype).cuda()
    ci = torch.zeros(2 * V, 3, 8 * S, 8 * S, dtype=dtype).cuda()
    out = p.get_noise_pred([lat[:V], lat[V:]], [emb[:V], emb[V:]], [ci[:V], ci[V:]], [ci[:V], ci[V:]], 321, 0.5, 0.25, 5.0)
    assert out.shape == (V, 4, S, S) and torch.isfinite(out).all()
    # oracle: un-tile, pair, zero residual rows for the reference image
    from test_unet import residuals
    x = lat.float().cpu().reshape(2 * V, 4, 2, S, S).permute(0, 2, 1, 3, 4).reshape(4 * V, 4, S, S)
    e = emb.float().cpu().unsqueeze(1).expand(-1, 2, -1, -1).reshape(4 * V, 77, 768)
    down, mid = residuals(cfg, 2 * V, S, seed=5)
    q = U.quantizer(dtype)
    zs = lambda r: torch.stack([torch.zeros_like(r), q(0.75 * r)], dim=1).view(-1, *r.shape[1:])
    with torch.no_grad():
        ref = U.unet_forward(sd, cfg, x, 321, e, 2, [zs(d) for d in down], zs(mid), q=q)
    ref = ref.view(2 * V, 2, 4, S, S)[:, 1]
    halves = ref
    ref = 5.0 * ref[V:] + (1 - 5.0) * ref[:V]
    bound = (5.0 + 4.0) * 2e-3 * max(halves[V:].norm(), halves[:V].norm()).item()
    err = (out.float().cpu() - ref).norm().item()
    assert err <= bound, (err, bound)


class _Nets:
    """Stand-in for diffusers' MultiControlNetModel: `.nets` + call; residuals = sum of per-net deterministic tensors."""

    def __init__(self, nets):
        self.nets = list(nets)

    def __call__(self, sample, t, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode, added_cond_kwargs,
                 return_dict):
        assert len(controlnet_cond) == len(self.nets) == len(conditioning_scale)
        down = mid = None
        for (cfg, seed), s in zip(self.nets, conditioning_scale):
            from test_unet import residuals
            d, m = residuals(cfg, 1, sample.shape[-1], seed=seed)
            # per-item scale taken from the item itself, so that the result does not depend on how views are chunked
            k = (1 + sample.float().mean(dim=(1, 2, 3))).view(-1, 1, 1, 1) * float(s)
            d, m = [(k * x.cuda()).to(sample.dtype) for x in d], (k * m.cuda()).to(sample.dtype)
            down, mid = (d, m) if down is None else ([a + b for a, b in zip(down, d)], mid + m)
        return down, mid


def test_two_pass_equals_one_pass(lib):
    """get_noise_pred_p1 caches the encoder; get_noise_pred_p2 adds the tile(+depth) ControlNet residuals and re-runs the
    decoder only (adapter3d_mixin.py:137-317).  ControlNet residuals enter only after the encoder (diffusers.py:110-121), so
    pass 2 must equal the 1-pass result with the same residual sum -- bitwise, same kernels on the same data."""
    cfg, dtype = U.TINY, torch.float16
    p, sd = _make(cfg, dtype, with_cn=False)
    p.controlnet = _Nets([(cfg, 11), (cfg, 12), (cfg, 13)])          # tile, depth, one extra
    V, S = 3, 16
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(2 * V, 4, S, S, generator=g).to(dtype).cuda()
    emb = torch.randn(2 * V, 77, 768, generator=g).to(dtype).cuda()
    img = torch.zeros(2 * V, 3, 8 * S, 8 * S, dtype=dtype).cuda()
    ch = lambda x: list(x.split(2, dim=0))
    for fuse in (True, False):
        p.fuse_chunks = fuse
        n1, dec_args, dec_kwargs = p.get_noise_pred_p1(ch(lat), ch(emb), 400, 6.0, ctrl_depths_batches=ch(img), depth_weight=0.3,
                                                       extra_control_batches=[ch(img)])
        assert n1.shape == (V, 4, S, S) and len(dec_args) == (1 if fuse else 3)
        n2 = p.get_noise_pred_p2(ch(lat), ch(emb), dec_args, dec_kwargs, 400, 6.0, ch(img), 0.7, ctrl_depths_batches=ch(img),
                                 depth_weight=0.3)
        # 1-pass with the same nets: tile 0.7 + depth 0.3 + extra 1.0 (pass 1) + depth 0.3 again (pass 2 re-applies nets[:2])
        one = _Nets([(cfg, 11), (cfg, 12), (cfg, 13), (cfg, 12)])
        down, mid = one(lat, 400, emb, [img] * 4, [0.7, 0.3, 1.0, 0.3], False, None, False)
        full = p.unet(lat, 400, encoder_hidden_states=emb, down_block_additional_residuals=down, mid_block_additional_residual=mid)[0]
        ref = 6.0 * full[V:].float() + (1 - 6.0) * full[:V].float()
        assert (n2.float() - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-3
        assert not torch.equal(n1, n2)
        n2b = p.get_noise_pred_p2(ch(lat), ch(emb), dec_args, dec_kwargs, 400, 6.0, ch(img), 0.7, ctrl_depths_batches=ch(img),
                                  depth_weight=0.3)
        assert torch.equal(n2, n2b), 'the cached encoder state must survive a decode'
        ad = p.get_noise_pred_p2(ch(lat), ch(emb), dec_args, dec_kwargs, 400, 6.0, ch(img), 0.7, adapter_scale=1.5)
        assert ad.shape == n2.shape


def test_two_pass_with_reference_latents(lib):
    """cond_noisy_latent_batches: write pass over the condition latents, then the sample pass reads ('r' in the encoder,
    'm' in the decoder so that pass 2 can decode again), adapter3d_mixin.py:193-224."""
    cfg, dtype = U.TINY, torch.float16
    p, sd = _make(cfg, dtype, with_cn=False)
    p.controlnet = _Nets([(cfg, 11), (cfg, 12)])
    V, S = 2, 16
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(2 * V, 4, S, S, generator=g).to(dtype)
    cond = torch.randn(2 * V, 4, S, S, generator=g).to(dtype)
    emb = torch.randn(2 * V, 77, 768, generator=g).to(dtype)
    img = torch.zeros(2 * V, 3, 8 * S, 8 * S, dtype=dtype).cuda()
    n1, dec_args, dec_kwargs = p.get_noise_pred_p1([lat.cuda()], [emb.cuda()], 400, 5.0, cond_noisy_latent_batches=[cond.cuda()])
    with torch.no_grad():
        q = U.quantizer(dtype)
        d_enc, d_dec = {}, {}
        e, r, x, c = U.unet_enc(sd, cfg, cond.float(), 400, emb.float(), q=q, attn_opts=dict(mode='w', ref_dict=d_enc))
        U.unet_dec(sd, cfg, e, r, x, emb.float(), q=q, attn_opts=dict(mode='w', ref_dict=d_dec))
        e, r, x, c = U.unet_enc(sd, cfg, lat.float(), 400, emb.float(), q=q, attn_opts=dict(mode='r', ref_dict=d_enc))
        out, _ = U.unet_dec(sd, cfg, e, r, x, emb.float(), q=q, attn_opts=dict(mode='m', ref_dict=d_dec))
        assert len(d_enc) == 0 and len(d_dec) > 0
    ref = 5.0 * out[V:] + (1 - 5.0) * out[:V]
    bound = (5.0 + 4.0) * 3e-3 * max(out[V:].norm(), out[:V].norm()).item()
    assert (n1.float().cpu() - ref).norm().item() <= bound
    # pass 2 decodes again with ControlNet residuals and still reads the decoder-side reference tokens
    n2 = p.get_noise_pred_p2([lat.cuda()], [emb.cuda()], dec_args, dec_kwargs, 400, 5.0, [img], 0.5)
    assert n2.shape == n1.shape and torch.isfinite(n2).all() and not torch.equal(n1, n2)

