"""Writes tests/golden/unet_encdec_ref.npz by EXECUTING the reference's in-tree `unet_enc` / `unet_dec`
(lib/models/architecture/diffusers.py:57-164, taken from the file with `ast`) over a stand-in UNet object whose blocks are the ORACLE's
block functions (oracle/unet_oracle.py: _resnet, _transformer, ...) arranged the way diffusers arranges them (CrossAttnDownBlock2D /
UNetMidBlock2DCrossAttn / CrossAttnUpBlock2D: skip tuples popped from the end).  What this pins is the in-tree part of SURVEY rows
a3-a4: the skip bookkeeping, the ControlNet residual additions (:113-121, :135-136), the slicing of the skips per up block, and the
hand-over between the two halves -- i.e. that the oracle's own unet_enc / unet_dec walk the blocks as the reference's functions do.
Run from the repo root (needs /root/reference):  python tests/golden/make_unet_encdec_golden.py"""
import ast
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import types
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import unet_oracle as U  # noqa: E402

REF = '/root/reference/lib/models/architecture/diffusers.py'
OUT = os.path.join(HERE, 'unet_encdec_ref.npz')


def stub_unet(sd, cfg):
    c = U._Ctx(sd, cfg, None, None)
    ch, L = cfg['block_out_channels'], cfg['layers_per_block']
    n = len(ch)
    nimg = lambda cak: (cak or {}).get('num_cross_attn_imgs', 1)

    class Down:
        def __init__(self, i):
            self.i, self.has_cross_attention = i, bool(cfg['down_attn'][i])

        def __call__(self, hidden_states, temb, encoder_hidden_states=None, cross_attention_kwargs=None):
            x, outs = hidden_states, ()
            for j in range(L):
                x = U._resnet(c, f'down_blocks.{self.i}.resnets.{j}', x, temb)
                if self.has_cross_attention:
                    x = U._transformer(c, f'down_blocks.{self.i}.attentions.{j}', x, encoder_hidden_states, cfg['num_heads'][self.i],
                                       cfg['transformer_layers'][self.i], nimg(cross_attention_kwargs))
                outs += (x,)
            if self.i < n - 1:
                x = U._conv(c, x, f'down_blocks.{self.i}.downsamplers.0.conv', stride=2)
                outs += (x,)
            return x, outs

    class Mid:
        has_cross_attention = True

        def __call__(self, sample, emb, encoder_hidden_states=None, cross_attention_kwargs=None):
            x = U._resnet(c, 'mid_block.resnets.0', sample, emb)
            x = U._transformer(c, 'mid_block.attentions.0', x, encoder_hidden_states, cfg['num_heads'][-1], cfg['transformer_layers'][-1],
                               nimg(cross_attention_kwargs))
            return U._resnet(c, 'mid_block.resnets.1', x, emb)

    class Up:
        def __init__(self, i):
            self.i, self.lvl = i, n - 1 - i
            self.has_cross_attention = bool(cfg['down_attn'][self.lvl])
            self.resnets = [None] * (L + 1)

        def __call__(self, hidden_states, temb, res_hidden_states_tuple, encoder_hidden_states=None, cross_attention_kwargs=None):
            x = hidden_states
            for j in range(L + 1):
                skip, res_hidden_states_tuple = res_hidden_states_tuple[-1], res_hidden_states_tuple[:-1]
                x = U._resnet(c, f'up_blocks.{self.i}.resnets.{j}', torch.cat([x, skip], dim=1), temb)
                if self.has_cross_attention:
                    x = U._transformer(c, f'up_blocks.{self.i}.attentions.{j}', x, encoder_hidden_states, cfg['num_heads'][self.lvl],
                                       cfg['transformer_layers'][self.lvl], nimg(cross_attention_kwargs))
            if self.i < n - 1:
                x = U._conv(c, F.interpolate(x, scale_factor=2.0, mode='nearest'), f'up_blocks.{self.i}.upsamplers.0.conv')
            return x

    u = types.SimpleNamespace(config=types.SimpleNamespace(center_input_sample=False), time_embed_act=None)
    u.get_time_embed = lambda sample, timestep: U.timestep_embedding(
        torch.as_tensor(timestep).reshape(-1).float().expand(sample.shape[0]), ch[0])
    u.time_embedding = lambda t: U._linear(c, F.silu(U._linear(c, t, 'time_embedding.linear_1')), 'time_embedding.linear_2')
    u.get_aug_embed = lambda emb, encoder_hidden_states, added_cond_kwargs: None
    u.process_encoder_hidden_states = lambda encoder_hidden_states, added_cond_kwargs: encoder_hidden_states
    u.conv_in = lambda x: U._conv(c, x, 'conv_in')
    u.down_blocks, u.mid_block, u.up_blocks = [Down(i) for i in range(n)], Mid(), [Up(i) for i in range(n)]
    u.conv_norm_out = lambda x: F.group_norm(x, cfg['norm_num_groups'], c.w('conv_norm_out.weight'), c.w('conv_norm_out.bias'), cfg['norm_eps'])
    u.conv_act, u.conv_out = F.silu, (lambda x: U._conv(c, x, 'conv_out'))
    return u


def case(seed=0):
    cfg = dict(U.TINY)
    sd = U.make_state_dict(cfg, seed=1234)
    g = torch.Generator().manual_seed(seed)
    B, S = 2, 8
    x = torch.randn(B, 4, S, S, generator=g)
    ctx = torch.randn(B, 11, cfg['cross_attention_dim'], generator=g)
    ch = cfg['block_out_channels']
    shapes = [(ch[0], S, S), (ch[0], S, S), (ch[0], S // 2, S // 2), (ch[1], S // 2, S // 2)]
    down = [0.3 * torch.randn(B, *s, generator=g) for s in shapes]
    mid = 0.3 * torch.randn(B, ch[1], S // 2, S // 2, generator=g)
    return cfg, sd, x, ctx, down, mid


def main():
    tree = ast.parse(open(REF).read())
    ns = dict(torch=torch, Union=Union, Optional=Optional, Dict=Dict, Any=Any, Tuple=Tuple)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ('unet_enc', 'unet_dec'):
            exec(compile(ast.Module([node], []), REF, 'exec'), ns)
    cfg, sd, x, ctx, down, mid = case()
    unet = stub_unet(sd, cfg)
    out = {}
    with torch.no_grad():
        for tag, cak, res in (('plain', None, (None, None)), ('controlnet', None, (down, mid)), ('paired', dict(num_cross_attn_imgs=2), (down, mid))):
            emb, skips, sample = ns['unet_enc'](unet, x, torch.tensor(321), ctx, cross_attention_kwargs=cak)
            y = ns['unet_dec'](unet, emb, skips, sample, ctx, cross_attention_kwargs=cak, down_block_additional_residuals=res[0],
                               mid_block_additional_residual=res[1])
            out[f'{tag}_emb'], out[f'{tag}_mid'], out[f'{tag}_out'] = emb.numpy(), sample.numpy(), y.numpy()
            # the skips are large: keep their shapes and two moments each (order matters: a swapped pair would show)
            out[f'{tag}_skip_shapes'] = np.array([list(s.shape) for s in skips])
            out[f'{tag}_skip_moments'] = np.array([[float(s.double().mean()), float(s.double().abs().mean())] for s in skips])
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), len(out), 'arrays')


if __name__ == '__main__':
    main()
