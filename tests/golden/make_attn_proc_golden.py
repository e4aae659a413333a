"""Writes tests/golden/attn_proc_ref.npz by EXECUTING the reference's attention processors on a stand-in `attn` module:
  AttnProcessor2_0, IPAttnProcessor2_0   lib/models/architecture/ip_adapter/attention_processor.py:184-396 (vendored)
  CrossImageAttnProcWrapper              lib/models/architecture/joint_attn.py:5-37
  ReferenceAttnProc                      lib/models/architecture/diffusers.py:646-673
  ReferenceOnlyAttnProc                  lib/pipelines/zero123plus.py:43-77
The class definitions are taken from the files with `ast` (the modules import diffusers).  `attn` carries what the processors touch of
diffusers' Attention: to_q / to_k / to_v (no bias), to_out = [Linear, Dropout], heads, and the switches that are off in the UNet
(spatial_norm, group_norm, norm_cross, residual_connection, rescale_output_factor = 1).
Run from the repo root (needs /root/reference):  python tests/golden/make_attn_proc_golden.py"""
import ast
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
from typing import Any

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
OUT = os.path.join(HERE, 'attn_proc_ref.npz')
C, HEADS, CTX, IP_TOKENS = 64, 4, 48, 4


def _cls(path, name, ns):
    tree = ast.parse(open(path).read())
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name][0]
    exec(compile(ast.Module([node], []), path, 'exec'), ns)
    return ns[name]


def weights(seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) / s[-1] ** 0.5
    return {'to_q.weight': r(C, C), 'self.to_k.weight': r(C, C), 'self.to_v.weight': r(C, C), 'cross.to_k.weight': r(C, CTX),
            'cross.to_v.weight': r(C, CTX), 'to_out.0.weight': r(C, C), 'to_out.0.bias': 0.1 * torch.randn(C, generator=g),
            'to_k_ip.weight': r(C, CTX), 'to_v_ip.weight': r(C, CTX)}


def make_attn(w, cross):
    a = nn.Module()
    a.heads, a.spatial_norm, a.group_norm, a.norm_cross, a.residual_connection, a.rescale_output_factor = HEADS, None, None, False, False, 1.0
    kind = 'cross' if cross else 'self'
    a.to_q, a.to_k, a.to_v = nn.Linear(C, C, bias=False), nn.Linear(CTX if cross else C, C, bias=False), nn.Linear(CTX if cross else C, C, bias=False)
    a.to_q.weight.data, a.to_k.weight.data, a.to_v.weight.data = w['to_q.weight'], w[f'{kind}.to_k.weight'], w[f'{kind}.to_v.weight']
    out = nn.Linear(C, C)
    out.weight.data, out.bias.data = w['to_out.0.weight'], w['to_out.0.bias']
    a.to_out = nn.ModuleList([out, nn.Dropout(0.0)])
    return a.eval()


def inputs(seed=1):
    g = torch.Generator().manual_seed(seed)
    return dict(x=torch.randn(4, 24, C, generator=g), ctx=torch.randn(4, 10 + IP_TOKENS, CTX, generator=g), ref=torch.randn(4, 24, C, generator=g))


def main():
    ns = dict(torch=torch, nn=nn, F=F, Any=Any, Attention=object)
    ap = os.path.join(REF, 'lib/models/architecture/ip_adapter/attention_processor.py')
    Attn2, IPAttn2 = _cls(ap, 'AttnProcessor2_0', ns), _cls(ap, 'IPAttnProcessor2_0', ns)
    Cross = _cls(os.path.join(REF, 'lib/models/architecture/joint_attn.py'), 'CrossImageAttnProcWrapper', ns)
    RefProc = _cls(os.path.join(REF, 'lib/models/architecture/diffusers.py'), 'ReferenceAttnProc', ns)
    RefOnly = _cls(os.path.join(REF, 'lib/pipelines/zero123plus.py'), 'ReferenceOnlyAttnProc', ns)
    w, io = weights(), inputs()
    x, ctx, ref = io['x'], io['ctx'], io['ref']
    sa, ca = make_attn(w, False), make_attn(w, True)
    out = {}
    with torch.no_grad():
        out['self'] = Attn2()(sa, x).numpy()
        out['cross'] = Attn2()(ca, x, encoder_hidden_states=ctx[:, :10]).numpy()
        out['cross_image_self'] = Cross(Attn2())(sa, x, num_cross_attn_imgs=2).numpy()
        out['cross_image_cross'] = Cross(Attn2())(ca, x, encoder_hidden_states=ctx[:, :10], num_cross_attn_imgs=2).numpy()
        ip = IPAttn2(hidden_size=C, cross_attention_dim=CTX, scale=0.7, num_tokens=IP_TOKENS)
        ip.to_k_ip.weight.data, ip.to_v_ip.weight.data = w['to_k_ip.weight'], w['to_v_ip.weight']
        out['ip'] = ip(ca, x, encoder_hidden_states=ctx).numpy()
        # reference attention (MVEdit's own processor): write pass over `ref`, then read ('r' pops) and keep ('m')
        proc = RefProc(Attn2(), enabled=True, name='layer')
        d = {}
        out['ref_w'] = proc(sa, ref, mode='w', ref_dict=d).numpy()
        out['ref_m'] = proc(sa, x, mode='m', ref_dict=d).numpy()
        out['ref_r'] = proc(sa, x, mode='r', ref_dict=d).numpy()
        assert not d
        # Zero123++'s processor with the classifier-free-guidance row kept out of the reference mechanism
        proc = RefOnly(Attn2(), enabled=True, name='layer')
        d = {}
        out['refonly_w'] = proc(sa, ref, mode='w', ref_dict=d, is_cfg_guidance=True).numpy()
        out['refonly_r'] = proc(sa, x, mode='r', ref_dict=d, is_cfg_guidance=True).numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
