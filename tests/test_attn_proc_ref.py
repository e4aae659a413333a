"""oracle/unet_oracle.py: _attention -- the single function through which the oracle evaluates every attention layer (plain, cross-image,
IP-Adapter, reference attention in its 'w' / 'r' / 'm' modes, with and without the classifier-free-guidance row) -- against the
REFERENCE's own processors executed on a stand-in `attn` module (tests/golden/attn_proc_ref.npz, written by
tests/golden/make_attn_proc_golden.py): AttnProcessor2_0 / IPAttnProcessor2_0 (vendored, attention_processor.py:184-396),
CrossImageAttnProcWrapper (joint_attn.py:5-37), ReferenceAttnProc (diffusers.py:646-673), ReferenceOnlyAttnProc (zero123plus.py:43-77).
This pins SURVEY section 8 rows a5-a8 of the UNet oracle; the block arithmetic around them (diffusers) stays unpinned."""
import importlib.util
import os

import numpy as np
import torch

from oracle import unet_oracle as U

HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, 'golden', 'attn_proc_ref.npz'))
spec = importlib.util.spec_from_file_location('make_attn_proc_golden', os.path.join(HERE, 'golden', 'make_attn_proc_golden.py'))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)


def _ctx(w, cross, **attn_opts):
    kind = 'cross' if cross else 'self'
    sd = {'a.to_q.weight': w['to_q.weight'], 'a.to_k.weight': w[f'{kind}.to_k.weight'], 'a.to_v.weight': w[f'{kind}.to_v.weight'],
          'a.to_out.0.weight': w['to_out.0.weight'], 'a.to_out.0.bias': w['to_out.0.bias'],
          'a.processor.to_k_ip.weight': w['to_k_ip.weight'], 'a.processor.to_v_ip.weight': w['to_v_ip.weight']}
    return U._Ctx(sd, {}, None, None, attn_opts)


def _close(got, key):
    np.testing.assert_allclose(got.numpy(), G[key], rtol=2e-5, atol=2e-6, err_msg=key)


def test_attention_branches_equal_reference_processors():
    w, io = M.weights(), M.inputs()
    x, ctx, ref = io['x'], io['ctx'], io['ref']
    H = M.HEADS
    with torch.no_grad():
        _close(U._attention(_ctx(w, False), 'a', x, None, H, 1), 'self')
        _close(U._attention(_ctx(w, True), 'a', x, ctx[:, :10], H, 1), 'cross')
        _close(U._attention(_ctx(w, False), 'a', x, None, H, 2), 'cross_image_self')
        _close(U._attention(_ctx(w, True), 'a', x, ctx[:, :10], H, 2), 'cross_image_cross')
        _close(U._attention(_ctx(w, True, ip_tokens=M.IP_TOKENS, ip_scale=0.7), 'a', x, ctx, H, 1), 'ip')
        d = {}
        _close(U._attention(_ctx(w, False, mode='w', ref_dict=d), 'a', ref, None, H, 1), 'ref_w')
        _close(U._attention(_ctx(w, False, mode='m', ref_dict=d), 'a', x, None, H, 1), 'ref_m')
        _close(U._attention(_ctx(w, False, mode='r', ref_dict=d), 'a', x, None, H, 1), 'ref_r')
        assert not d                                                   # 'r' pops, as the reference's processor does
        d = {}
        _close(U._attention(_ctx(w, False, mode='w', ref_dict=d, ref_skip=1), 'a', ref, None, H, 1), 'refonly_w')
        _close(U._attention(_ctx(w, False, mode='r', ref_dict=d, ref_skip=1), 'a', x, None, H, 1), 'refonly_r')
