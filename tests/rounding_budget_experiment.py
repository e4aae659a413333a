"""Which storage roundings carry the UNet's end-to-end 16-bit error: the oracle re-walked with one quantizer PER
ROUNDING SITE, so that single sites can be switched to fp32 and the engine's own rounding points (fused epilogues: conv1 + time embedding,
conv2 + shortcut, the residual adds) can be emulated instead of PyTorch's.  Measured against pure fp32 arithmetic over the same 16-bit
weights at the benchmark latent size.  CPU only (torch); a few minutes.  Test infrastructure: uses the oracle.

Sites:  norm   GroupNorm(+SiLU) / LayerNorm outputs            mm     conv / linear outputs inside a branch
        attn   SDPA output                                      act    GEGLU hidden
        res    the residual stream x + f(x) (hi + lo pair in the engine's `residual_fp32` mode = unrounded)
        bin    what a branch READS of the stream (the engine's norms read the 16-bit `hi` half only = rounded, unless MVE reads the pair)
Options: --bf16 (also bf16), --phase (Upsample2D as four 2 x 2 phase convs with summed, once-rounded weights), --lo8 (the stream pair with the 8-bit
low half the engine stores since round 5 against an exact low half).
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_oracle as UO
from mvedit_amd.unet import SD15_CONFIG

ident = lambda v: v


class Sites:
    def __init__(self, q, **off):
        for k in ('norm', 'mm', 'attn', 'act', 'res', 'bin'):
            setattr(self, k, ident if off.get(k) else q)
        self.where = ''          # the block being walked (site selectors of derived experiments look at it)
        if off.get('lo8'):       # the stream as the engine stores it since round 5: hi = q(v) plus an 8-bit low half E5M2(2^8 (v - hi)); norms read hi + lo
            def pair8(v):
                hi = q(v)
                return hi + ((v - hi) * 256.0).to(torch.float8_e5m2).float() / 256.0
            self.res, self.bin = pair8, ident
            if off.get('h1pair'):
                self.h1 = pair8


upsample_conv_phases = UO.upsample_conv_phases


def run(sd, cfg, x, t, ctx, q, engine=True, phase_ups=False, **off):
    """engine=True: the engine's rounding points (one rounding per fused op); False: PyTorch-half's (every op output).
    phase_ups: the upsampler convs as four 2x2 phase convs whose summed weights are rounded to the storage type."""
    S = Sites(q, **off)
    c = UO._Ctx(sd, cfg, q, None)

    def conv(h, name, stride=1, padding=1, qq=None):
        S.where = name
        return (qq or S.mm)(F.conv2d(h, c.w(name + '.weight'), c.w(name + '.bias'), stride=stride, padding=padding))

    def linear(h, name, bias=True, qq=None):
        return (qq or S.mm)(F.linear(h, c.w(name + '.weight'), c.w(name + '.bias') if bias else None))

    def resnet(p, xs, temb):
        g, eps = cfg['norm_num_groups'], cfg['norm_eps']
        S.where = p           # (site selectors of derived experiments look at the block they are in)
        xin = S.bin(xs)
        h = S.norm(F.silu(F.group_norm(xin, g, c.w(p + '.norm1.weight'), c.w(p + '.norm1.bias'), eps)))
        tt = F.linear(q(F.silu(temb)), c.w(p + '.time_emb_proj.weight'), c.w(p + '.time_emb_proj.bias'))       # fp32 row vector in the engine
        if engine:
            # conv1's output is read by GroupNorm only -- never as an MFMA operand: `h1` = the site on its own (a pair there costs one byte per element)
            h = getattr(S, 'h1', S.mm)(F.conv2d(h, c.w(p + '.conv1.weight'), c.w(p + '.conv1.bias'), padding=1) + tt[:, :, None, None])
        else:
            h = S.mm(conv(h, p + '.conv1') + S.mm(tt)[:, :, None, None])
        h = S.norm(F.silu(F.group_norm(h, g, c.w(p + '.norm2.weight'), c.w(p + '.norm2.bias'), eps)))
        h = F.conv2d(h, c.w(p + '.conv2.weight'), c.w(p + '.conv2.bias'), padding=1)
        if (p + '.conv_shortcut.weight') in sd:
            sc = F.conv2d(q(xs), c.w(p + '.conv_shortcut.weight'), c.w(p + '.conv_shortcut.bias'))                # MFMA operand: always the 16-bit half
            return S.res(h + sc) if engine else S.res(S.mm(sc) + S.mm(h))
        return S.res(xs + h) if engine else S.res(xs + S.mm(h))

    def sdpa(qv, k, v, heads):
        B, _, C = qv.shape
        d = C // heads
        qv, k, v = (z.view(B, -1, heads, d).transpose(1, 2) for z in (qv, k, v))
        a = torch.softmax(qv @ k.transpose(-1, -2) * (d ** -0.5), dim=-1) @ v
        return S.attn(a.transpose(1, 2).reshape(B, -1, C))

    def attention(p, n, cx, heads):
        kv = n if cx is None else cx
        a = sdpa(linear(n, p + '.to_q', False), linear(kv, p + '.to_k', False), linear(kv, p + '.to_v', False), heads)
        return F.linear(a, c.w(p + '.to_out.0.weight'), c.w(p + '.to_out.0.bias'))          # unrounded: the caller adds the residual first (engine) or rounds (torch)

    def transformer(p, xs, cx, heads, layers):
        B, C, H, W = xs.shape
        S.where = p
        h = S.norm(F.group_norm(S.bin(xs), cfg['norm_num_groups'], c.w(p + '.norm.weight'), c.w(p + '.norm.bias'), 1e-6))
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = S.res(F.linear(h, c.w(p + '.proj_in.weight').flatten(1), c.w(p + '.proj_in.bias')))          # starts the block's stream
        r = (lambda v: v) if engine else S.mm
        for k in range(layers):
            b = f'{p}.transformer_blocks.{k}'
            n = S.norm(F.layer_norm(S.bin(h), (C,), c.w(b + '.norm1.weight'), c.w(b + '.norm1.bias'), 1e-5))
            h = S.res(h + r(attention(b + '.attn1', n, None, heads)))
            n = S.norm(F.layer_norm(S.bin(h), (C,), c.w(b + '.norm2.weight'), c.w(b + '.norm2.bias'), 1e-5))
            h = S.res(h + r(attention(b + '.attn2', n, cx, heads)))
            n = S.norm(F.layer_norm(S.bin(h), (C,), c.w(b + '.norm3.weight'), c.w(b + '.norm3.bias'), 1e-5))
            val, gate = F.linear(n, c.w(b + '.ff.net.0.proj.weight'), c.w(b + '.ff.net.0.proj.bias')).chunk(2, dim=-1)
            f = S.act(val * F.gelu(gate))
            h = S.res(h + r(F.linear(f, c.w(b + '.ff.net.2.weight'), c.w(b + '.ff.net.2.bias'))))
        # proj_out reads the stream as an MFMA operand: the 16-bit half
        o = F.linear(q(h), c.w(p + '.proj_out.weight').flatten(1), c.w(p + '.proj_out.bias'))
        o = o.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return S.res(r(o) + xs)

    ch = cfg['block_out_channels']
    n_lv = len(ch)
    sample, cx = q(x), q(ctx)
    tt = t.reshape(-1).float().expand(sample.shape[0])
    emb = S.mm(F.linear(q(UO.timestep_embedding(tt, ch[0])), c.w('time_embedding.linear_1.weight'), c.w('time_embedding.linear_1.bias')))
    emb = S.mm(F.linear(q(F.silu(emb)), c.w('time_embedding.linear_2.weight'), c.w('time_embedding.linear_2.bias')))
    h = conv(sample, 'conv_in', qq=S.res)
    res = [h]
    # a stream tensor consumed as an MFMA operand (down / upsampler convs, conv_shortcut, skip concat) is the 16-bit half
    op = q        # (idempotent on an already rounded stream)
    for i in range(n_lv):
        for j in range(cfg['layers_per_block']):
            h = resnet(f'down_blocks.{i}.resnets.{j}', h, emb)
            if cfg['down_attn'][i]:
                h = transformer(f'down_blocks.{i}.attentions.{j}', h, cx, cfg['num_heads'][i], cfg['transformer_layers'][i])
            res.append(h)
        if i < n_lv - 1:
            h = conv(op(h), f'down_blocks.{i}.downsamplers.0.conv', stride=2, qq=S.res)
            res.append(h)
    h = resnet('mid_block.resnets.0', h, emb)
    h = transformer('mid_block.attentions.0', h, cx, cfg['num_heads'][-1], cfg['transformer_layers'][-1])
    h = resnet('mid_block.resnets.1', h, emb)
    rev = lambda k: list(reversed(cfg[k]))
    for i in range(n_lv):
        for j in range(cfg['layers_per_block'] + 1):
            h = torch.cat([h, res.pop()], dim=1)
            h = resnet(f'up_blocks.{i}.resnets.{j}', h, emb)
            if rev('down_attn')[i]:
                h = transformer(f'up_blocks.{i}.attentions.{j}', h, cx, rev('num_heads')[i], rev('transformer_layers')[i])
        if i < n_lv - 1:
            if phase_ups:
                nm = f'up_blocks.{i}.upsamplers.0.conv'
                S.where = nm
                h = S.res(upsample_conv_phases(op(h), c.w(nm + '.weight'), c.w(nm + '.bias'), q))
            else:
                h = conv(F.interpolate(op(h), scale_factor=2.0, mode='nearest'), f'up_blocks.{i}.upsamplers.0.conv', qq=S.res)
    S.where = 'conv_norm_out'
    h = S.norm(F.silu(F.group_norm(S.bin(h), cfg['norm_num_groups'], c.w('conv_norm_out.weight'), c.w('conv_norm_out.bias'), cfg['norm_eps'])))
    return F.conv2d(h, c.w('conv_out.weight'), c.w('conv_out.bias'), padding=1)


if __name__ == '__main__':
    cfg = SD15_CONFIG
    torch.set_num_threads(os.cpu_count() or 1)
    dts = [torch.float16] + ([torch.bfloat16] if '--bf16' in sys.argv else [])
    for dt in dts:
        sd = {k: v.to(dt).float() for k, v in UO.make_state_dict(cfg, seed=1234).items()}
        g = torch.Generator().manual_seed(0)
        x = torch.randn(1, 4, 64, 64, generator=g).to(dt).float()
        ctx = torch.randn(1, 77, 768, generator=g).to(dt).float()
        t = torch.tensor([499.0])
        q = UO.quantizer(dt)
        with torch.no_grad():
            ref = UO.unet_forward(sd, cfg, x, t, ctx)
            rel = lambda a: float((a - ref).norm() / ref.norm())
            chk = run(sd, cfg, x, t, ctx, ident)
            print(f'{dt}: re-walk with no rounding vs oracle fp32: {rel(chk):.2e} (must be ~1e-6)', flush=True)
            print(f'  oracle, every op output rounded (PyTorch half)          {rel(UO.unet_forward(sd, cfg, x, t, ctx, q=q)):.3e}', flush=True)
            if '--phase' in sys.argv:
                xx = torch.randn(2, 8, 5, 6)
                ww, bb = torch.randn(4, 8, 3, 3), torch.randn(4)
                d = (upsample_conv_phases(xx, ww, bb) - F.conv2d(F.interpolate(xx, scale_factor=2.0), ww, bb, padding=1)).abs().max()
                print(f'  phase decomposition vs upsample + conv, fp32: max |diff| {float(d):.1e}')
                for label, off in [('engine', {}), ('engine + pair + pair-reading norms', dict(res=1, bin=1))]:
                    a = rel(run(sd, cfg, x, t, ctx, q, True, **off))
                    b = rel(run(sd, cfg, x, t, ctx, q, True, phase_ups=True, **off))
                    print(f'  {label:40s} 3x3 on the upsampled input {a:.3e}   four 2x2 phase convs, summed weights rounded {b:.3e}', flush=True)
                continue
            if '--lo8' in sys.argv:
                for label, off in [('engine + stream pair (fp32-exact low half), pair-reading norms, phase upsamplers', dict(res=1, bin=1)),
                                   ('engine + stream pair with the 8-bit low half (lo8: what ships since round 5), phase upsamplers', dict(lo8=1))]:
                    print(f'  {label:100s} {rel(run(sd, cfg, x, t, ctx, q, True, phase_ups=True, **off)):.3e}', flush=True)
                continue
            for label, eng, off in [
                ('PyTorch-half rounding points (re-walk)', False, {}),
                ('engine rounding points', True, {}),
                ('engine + stream pair (branches read hi)', True, dict(res=1)),
                ('engine + stream pair, norms read the pair', True, dict(res=1, bin=1)),
                ('engine, norm outputs fp32', True, dict(norm=1)),
                ('engine, mm outputs fp32', True, dict(mm=1)),
                ('engine, SDPA output fp32', True, dict(attn=1)),
                ('engine, GEGLU hidden fp32', True, dict(act=1)),
                ('engine + pair + pair-reading norms + norm outputs fp32', True, dict(res=1, bin=1, norm=1)),
                ('engine + pair + pair-reading norms + mm fp32', True, dict(res=1, bin=1, mm=1)),
            ]:
                print(f'  {label:58s} {rel(run(sd, cfg, x, t, ctx, q, eng, **off)):.3e}', flush=True)
