"""`Adapter3DMixin.get_noise_pred` orchestration (chunk walk, reference pairing, ControlNet residual padding, CFG / adapter-scale
combine) against the output of the REFERENCE's own method executed over the same deterministic stand-in networks
(tests/golden/mixin_ref.npz, written by tests/golden/make_mixin_golden.py from lib/pipelines/adapter3d_mixin.py:68-135).
Host logic only: runs on the CPU with torch stand-ins in place of the engines; the native engines behind the same mirror are covered
by tests/test_pipeline_mixin.py on the GPU."""
import os

import numpy as np
import pytest
import torch

import stubs
from mvedit_amd.pipelines import Adapter3DMixin

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mixin_ref.npz'))


class Pipe(Adapter3DMixin):
    def __init__(self, fuse):
        self.unet, self.controlnet, self.fuse_chunks = stubs.StubUNet(), stubs.StubControlNet(), fuse


@pytest.mark.parametrize('fuse', [False, True])
@pytest.mark.parametrize('name', ['plain', 'no_depth_adapter_scale', 'paired'])
def test_get_noise_pred_equals_reference_output(name, fuse):
    """fuse=False is the reference's chunk walk; fuse=True concatenates the chunks into one UNet batch (the MI355X-first default):
    both must reproduce what the reference's method returned."""
    kw = stubs.cases()[name]
    with torch.no_grad():
        out = Pipe(fuse).get_noise_pred(**kw)
    np.testing.assert_allclose(out.numpy(), G[name], rtol=1e-5, atol=1e-6)


class Pipe2(Adapter3DMixin):
    def __init__(self, fuse, n_nets):
        self.unet, self.fuse_chunks = object(), fuse
        self.controlnet = stubs.StubMulti([stubs.StubNet(k) for k in range(n_nets)])
        self.negative_prompt_embeds = torch.linspace(-1, 1, 7 * 16).view(1, 7, 16)


@pytest.mark.parametrize('fuse', [False, True])
@pytest.mark.parametrize('name', ['plain', 'paired', 'reference_attention_no_depth'])
def test_two_pass_equals_reference_output(monkeypatch, name, fuse):
    """get_noise_pred_p1 / _p2 (lib/pipelines/adapter3d_mixin.py:137-317): which ControlNets run in which pass, the cached encoder state,
    the residual sums of pass 2, reference attention through cond_noisy_latent_batches, ctrl_text_embedding=False, adapter_scale --
    against the reference's own two methods executed over the same stand-ins (unet_enc / unet_dec / MultiControlNetModel replaced alike
    on both sides)."""
    import mvedit_amd.pipelines.adapter3d_mixin as M
    monkeypatch.setattr(M, 'unet_enc', stubs.stub_unet_enc)
    monkeypatch.setattr(M, 'unet_dec', stubs.stub_unet_dec)
    kw = stubs.cases_2pass()[name]
    pipe = Pipe2(fuse, 2 + len(kw['p1'].get('extra_control_batches') or []))
    with torch.no_grad():
        noise1, dec_args, dec_kwargs = pipe.get_noise_pred_p1(**kw['p1'])
        noise2 = pipe.get_noise_pred_p2(dec_args=dec_args, dec_kwargs=dec_kwargs, **kw['p2'])
    np.testing.assert_allclose(noise1.numpy(), G[f'2pass_{name}_p1'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(noise2.numpy(), G[f'2pass_{name}_p2'], rtol=1e-5, atol=1e-6)


class SharingNet(stubs.StubNet):
    shares_cond = True           # like mvedit_amd.controlnet.ControlNetEngine: fewer conditioning images than batch items are tiled over the batch


class SharingMulti(stubs.StubMulti):
    """StubMulti whose nets accept B / R conditioning images (item b uses image b mod len(cond)); records the conditioning batch sizes it saw."""
    seen = []

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, **kw):
        b = sample.shape[0]
        SharingMulti.seen.append([c.shape[0] for c in controlnet_cond])
        tiled = [c if c.shape[0] == b else c.repeat(b // c.shape[0], 1, 1, 1) for c in controlnet_cond]
        return super().__call__(sample, timestep, encoder_hidden_states, tiled, conditioning_scale, **kw)


def test_shared_control_images_of_the_cfg_halves_keep_the_reference_output(monkeypatch):
    """The reference builds the control-image lists of the two CFG halves from the same tensors (`x.split(diff_bs) * 2`,
    mvedit_3d_pipeline.py:1232 / :1417).  With ControlNets that accept shared images the fused walk hands over ONE half (object identity of the
    list halves, no data comparison) -- in the 2-pass methods too -- and the noise predictions stay the reference's own (golden outputs of
    lib/pipelines/adapter3d_mixin.py over the same stand-ins).  Lists whose halves are different objects are passed whole."""
    import mvedit_amd.pipelines.adapter3d_mixin as M
    monkeypatch.setattr(M, 'unet_enc', stubs.stub_unet_enc)
    monkeypatch.setattr(M, 'unet_dec', stubs.stub_unet_dec)
    kw = stubs.cases_2pass()['plain']
    V = 4
    halves = lambda batches: tuple(batches[:len(batches) // 2]) * 2          # same data as the case (V even: the chunks do not straddle the halves)
    p1, p2 = dict(kw['p1']), dict(kw['p2'])
    for d, keys in ((p1, ('ctrl_depths_batches',)), (p2, ('ctrl_images_batches', 'ctrl_depths_batches'))):
        for k in keys:
            assert all(torch.equal(a, b) for a, b in zip(halves(d[k]), d[k]))
            d[k] = halves(d[k])
    p1['extra_control_batches'] = [halves(e) for e in p1['extra_control_batches']]
    pipe = Pipe2(True, 3)
    pipe.controlnet = SharingMulti([SharingNet(k) for k in range(3)])
    SharingMulti.seen = []
    with torch.no_grad():
        noise1, dec_args, dec_kwargs = pipe.get_noise_pred_p1(**p1)
        noise2 = pipe.get_noise_pred_p2(dec_args=dec_args, dec_kwargs=dec_kwargs, **p2)
    np.testing.assert_allclose(noise1.numpy(), G['2pass_plain_p1'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(noise2.numpy(), G['2pass_plain_p2'], rtol=1e-5, atol=1e-6)
    assert SharingMulti.seen == [[V, V], [V, V]]                             # pass 1: depth + extra; pass 2: tile + depth -- one half each
    # the case as committed: `torch.cat([x] * 2).split(diff_bs)` -- the form of the reference's ordinary 1-pass branch (mvedit_3d_pipeline.py:1238-1241):
    # different objects, consecutive views of one tensor with equal halves.  Round 5: recognised (the chunks viewed as that tensor again + one
    # comparison of its halves), unless the comparison is switched off; chunks that own their storage are never assumed to repeat
    SharingMulti.seen = []
    with torch.no_grad():
        noise1, dec_args, dec_kwargs = pipe.get_noise_pred_p1(**kw['p1'])
        noise2 = pipe.get_noise_pred_p2(dec_args=dec_args, dec_kwargs=dec_kwargs, **kw['p2'])
    assert SharingMulti.seen == [[V, V], [V, V]]
    np.testing.assert_allclose(noise1.numpy(), G['2pass_plain_p1'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(noise2.numpy(), G['2pass_plain_p2'], rtol=1e-5, atol=1e-6)
    for variant in ('off', 'clones', 'halves_differ'):
        q1, q2 = dict(kw['p1']), dict(kw['p2'])
        if variant == 'clones':
            for d, keys in ((q1, ('ctrl_depths_batches',)), (q2, ('ctrl_images_batches', 'ctrl_depths_batches'))):
                for k in keys:
                    d[k] = tuple(b.clone() for b in d[k])
            q1['extra_control_batches'] = [tuple(b.clone() for b in e) for e in q1['extra_control_batches']]
        if variant == 'halves_differ':
            x = torch.cat(list(q2['ctrl_images_batches'])).clone()
            x[-1] += 1.0
            q2['ctrl_images_batches'] = x.split(q2['ctrl_images_batches'][0].shape[0])
        pipe.detect_repeated_cond = variant != 'off'
        SharingMulti.seen = []
        with torch.no_grad():
            n1, da, dk = pipe.get_noise_pred_p1(**q1)
            pipe.get_noise_pred_p2(dec_args=da, dec_kwargs=dk, **q2)
        pipe.detect_repeated_cond = True
        assert SharingMulti.seen == ([[V, V], [2 * V, V]] if variant == 'halves_differ' else [[2 * V, 2 * V], [2 * V, 2 * V]]), (variant, SharingMulti.seen)
        np.testing.assert_allclose(n1.numpy(), G['2pass_plain_p1'], rtol=1e-5, atol=1e-6)


def test_chunks_viewed_as_one_tensor():
    """Adapter3DMixin._as_one_tensor: `x.split(n)` chunks (ragged last chunk, a chunk straddling the CFG halves) give x back without a copy; anything
    else -- reordered, cloned, strided, mixed dtypes -- gives None."""
    from mvedit_amd.pipelines import Adapter3DMixin as A
    x = torch.arange(64 * 3 * 2 * 2, dtype=torch.float32).reshape(64, 3, 2, 2)
    for n in (6, 8, 64, 5):
        y = A._as_one_tensor(list(x.split(n)))
        assert y is not None and y.data_ptr() == x.data_ptr() and torch.equal(y, x)
    ch = list(x.split(6))
    assert A._as_one_tensor(ch[1:]) is not None and torch.equal(A._as_one_tensor(ch[1:]), x[6:])
    assert A._as_one_tensor([ch[1], ch[0]]) is None
    assert A._as_one_tensor([ch[0], ch[1].clone()]) is None
    assert A._as_one_tensor([ch[0], ch[2]]) is None                        # a gap
    assert A._as_one_tensor(list(x[:, :, :1].split(6))) is None            # not contiguous
    assert A._as_one_tensor([ch[0], ch[1].double()]) is None
    two = torch.cat([x[:32]] * 2)
    assert torch.equal(A()._cat_shared_cond(list(two.split(6))), x[:32])   # 11 chunks, the sixth straddles the halves
    assert A()._cat_shared_cond(list(x.split(6))).shape[0] == 64           # halves differ: whole
