"""Tri-plane decoders (csrc/triplane.hip behind mvedit_amd.triplane) vs the reference's own `TriPlaneDecoder.point_decode` /
`TriPlaneiNGPDecoder.point_decode` EXECUTED (tests/golden/triplane_ref.npz, tests/golden/make_triplane_golden.py).
Bars: fp32 MLPs of width 128 over O(1) activations -> 2e-5 relative on sigma (an exponential), 2e-5 absolute on rgb."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import triplane_oracle as TO

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'triplane_ref.npz'))
CASES = {'plain': dict(plane_cfg=('xy', 'xz', 'yz'), flip_z=False, activation='silu', ingp=False),
         'flip': dict(plane_cfg=('xy', 'yz', 'xz'), flip_z=True, activation='relu', ingp=False),
         'ingp': dict(plane_cfg=('xy', 'xz', 'yz'), flip_z=False, activation='silu', ingp=True)}
HASH = dict(n_levels=12, max_resolution=320, log2_hashmap_size=12, bound=1.0)


def weights(tag, dtype=torch.float32):
    keys = ['base_w', 'base_b', 'dens_w', 'dens_b', 'col1_w', 'col1_b', 'col2_w', 'col2_b'] + (['ingp_w', 'ingp_b'] if CASES[tag]['ingp'] else [])
    return {k: torch.from_numpy(G[f'{tag}_{k}']).to(dtype) for k in keys}


@pytest.mark.parametrize('tag', list(CASES))
def test_oracle_restatement_equals_reference_output(tag):
    c = CASES[tag]
    hash_ = dict(HASH, table=G[f'{tag}_table']) if c['ingp'] else None
    t = lambda k: torch.from_numpy(G[f'{tag}_{k}'])
    sig, rgb = TO.point_decode(t('xyz'), t('dirs'), t('code')[0], weights(tag), c['plane_cfg'], c['flip_z'], c['activation'], hash=hash_)
    np.testing.assert_allclose(sig.numpy(), G[f'{tag}_sigmas'], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(rgb.numpy(), G[f'{tag}_rgbs'], rtol=0, atol=2e-6)
    assert G[f'{tag}_sigmas'].std() > 0.05 and G[f'{tag}_rgbs'].std() > 0.02
    assert (np.abs(G[f'{tag}_xyz']) > 1).any(), 'border padding must be exercised'


def test_descriptor_layout_matches_the_header(tmp_path):
    pytest.importorskip('mvedit_amd._lib')
    from mvedit_amd.triplane import _Desc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f[0] for f in _Desc._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mvedit_amd.h"\nint main(void) {\n  printf("%zu", sizeof(MveTriplaneDesc));\n'
                   + ''.join(f'  printf(" %zu", offsetof(MveTriplaneDesc, {f}));\n' for f in fields) + '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)], check=True)
    nums = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(_Desc) and nums[1:] == [getattr(_Desc, f).offset for f in fields]


def _engine(tag):
    from mvedit_amd.triplane import TriPlaneDecoder, TriPlaneiNGPDecoder
    c = CASES[tag]
    kw = dict(plane_cfg=c['plane_cfg'], activation=c['activation'], flip_z=c['flip_z'])
    eng = TriPlaneiNGPDecoder(n_levels=12, max_resolution=320, log2_hashmap_size=12, **kw) if c['ingp'] else TriPlaneDecoder(**kw)
    sd = {'base_net.0.weight': G[f'{tag}_base_w'], 'base_net.0.bias': G[f'{tag}_base_b'], 'density_net.0.weight': G[f'{tag}_dens_w'],
          'density_net.0.bias': G[f'{tag}_dens_b'], 'color_net.0.weight': G[f'{tag}_col1_w'], 'color_net.0.bias': G[f'{tag}_col1_b'],
          'color_net.2.weight': G[f'{tag}_col2_w'], 'color_net.2.bias': G[f'{tag}_col2_b']}
    if c['ingp']:
        sd.update({'ingp_base_net.0.weight': G[f'{tag}_ingp_w'], 'ingp_base_net.0.bias': G[f'{tag}_ingp_b'], 'encoder.params': G[f'{tag}_table'].reshape(-1)})
    return eng.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})


def test_packed_parameter_copies_follow_edits():
    """the kernel-side copies (_pack) are rebuilt when torch's version counter, the tensor's identity or its storage address move;
    `.data` edits move none of them -- `invalidate()` is the documented way -- and every packed tensor is a copy, so a missed edit leaves the set
    consistent (all stale) instead of fresh aliases next to stale transposes.  No GPU needed: packing is host-side bookkeeping."""
    pytest.importorskip('mvedit_amd._lib')
    from mvedit_amd.triplane import TriPlaneDecoder
    tag = next(t for t in CASES if not CASES[t]['ingp'])
    c = CASES[tag]
    eng = TriPlaneDecoder(plane_cfg=c['plane_cfg'], activation=c['activation'], flip_z=c['flip_z'], device='cpu')
    sd = {'base_net.0.weight': G[f'{tag}_base_w'], 'base_net.0.bias': G[f'{tag}_base_b'], 'density_net.0.weight': G[f'{tag}_dens_w'],
          'density_net.0.bias': G[f'{tag}_dens_b'], 'color_net.0.weight': G[f'{tag}_col1_w'], 'color_net.0.bias': G[f'{tag}_col1_b'],
          'color_net.2.weight': G[f'{tag}_col2_w'], 'color_net.2.bias': G[f'{tag}_col2_b']}
    eng.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    p_b, p_w = eng.params['base_net.0.bias'], eng.params['base_net.0.weight']
    assert eng.w['base_b'].data_ptr() != p_b.data_ptr()                      # a copy, not an alias
    p_b.add_(1.0)                                                            # in-place op: version counter moves
    eng._pack()
    assert torch.equal(eng.w['base_b'], p_b)
    p_b.data.mul_(2.0)                                                       # through .data: nothing moves ...
    p_w.data.mul_(2.0)
    eng._pack()
    assert not torch.equal(eng.w['base_b'], p_b) and not torch.equal(eng.w['base_wT'], p_w.t())      # ... both copies stale, consistently
    eng.invalidate()
    eng._pack()
    assert torch.equal(eng.w['base_b'], p_b) and torch.equal(eng.w['base_wT'], p_w.t())


@pytest.mark.gpu
@pytest.mark.parametrize('tag', list(CASES))
def test_hip_vs_reference_output(lib, tag):
    eng = _engine(tag)
    t = lambda k: torch.from_numpy(G[f'{tag}_{k}']).cuda()
    sig, rgb, n = eng.point_decode([t('xyz')], [t('dirs')], t('code'))
    assert n == [G[f'{tag}_xyz'].shape[0]]
    np.testing.assert_allclose(sig.cpu().numpy(), G[f'{tag}_sigmas'], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(rgb.cpu().numpy(), G[f'{tag}_rgbs'], rtol=0, atol=2e-5)
    sig2, n2 = eng.point_density_decode([t('xyz')], t('code'))
    assert torch.equal(sig2, sig) and n2 == n
    with pytest.raises(NotImplementedError):
        eng.point_decode([t('xyz').requires_grad_(True)], [t('dirs')], t('code'))


@pytest.mark.gpu
def test_hip_large_batch_vs_oracle_and_timing(lib):
    """2^18 points of a 128^2 x 32-channel tri-plane (the SSDNeRF code size): vs the float64 oracle; prints points/s."""
    tag = 'plain'
    eng = _engine(tag)
    g = torch.Generator().manual_seed(5)
    N = 1 << 18
    xyz = torch.rand(N, 3, generator=g) * 2.2 - 1.1
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    code = torch.randn(1, 3, 32, 128, 128, generator=g)
    sig, rgb, _ = eng.point_decode([xyz.cuda()], [dirs.cuda()], code.cuda())
    idx = torch.randperm(N, generator=g)[:4096]
    c = CASES[tag]
    so, ro = TO.point_decode(xyz[idx].double(), dirs[idx].double(), code[0].double(), weights(tag, torch.float64), c['plane_cfg'], c['flip_z'], c['activation'])
    np.testing.assert_allclose(sig.cpu()[idx].numpy(), so.numpy(), rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(rgb.cpu()[idx].numpy(), ro.numpy(), rtol=0, atol=2e-5)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xg, dg, cg = xyz.cuda(), dirs.cuda(), code.cuda()
    eng.point_decode([xg], [dg], cg)
    a.record()
    for _ in range(5):
        eng.point_decode([xg], [dg], cg)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f'tri-plane decode: {N} points in {ms:.3f} ms = {N / ms / 1e6:.2f} G points/s, {N * 31.2e3 * 2 / ms / 1e9:.1f} TFLOP/s fp32')


_ONAME = {'base_net.0.weight': 'base_w', 'base_net.0.bias': 'base_b', 'density_net.0.weight': 'dens_w', 'density_net.0.bias': 'dens_b',
          'color_net.0.weight': 'col1_w', 'color_net.0.bias': 'col1_b', 'color_net.2.weight': 'col2_w', 'color_net.2.bias': 'col2_b',
          'ingp_base_net.0.weight': 'ingp_w', 'ingp_base_net.0.bias': 'ingp_b'}


@pytest.mark.gpu
@pytest.mark.parametrize('density_only', [False, True], ids=['sigma+rgb', 'sigma'])
@pytest.mark.parametrize('tag', list(CASES))
def test_hip_backward_vs_oracle_autograd(lib, tag, density_only):
    """mve_triplane_backward (behind point_decode_autograd) against torch autograd over the float64 oracle: gradients of a random linear
    functional of (sigma, rgb) w.r.t. the code planes and every Linear; the hash-table gradient (the oracle's encoder is numpy) through a
    directional finite difference of the oracle along a random table direction (sigma and rgb are smooth in the table)."""
    c = CASES[tag]
    eng = _engine(tag)
    g = torch.Generator().manual_seed(11)
    N = 3001                                                              # odd: every workspace section behind ws_do starts off a 16-byte boundary for C = 6
    xyz = torch.rand(N, 3, generator=g) * 2.2 - 1.1                       # some points outside the planes: border padding
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    code = torch.from_numpy(G[f'{tag}_code']).clone()
    g_sig = torch.randn(N, generator=g) * 0.1
    g_rgb = torch.randn(N, 3, generator=g)

    def oracle_loss(w64, code64, table):
        hash_ = dict(HASH, table=table) if c['ingp'] else None
        so, ro = TO.point_decode(xyz.double(), None if density_only else dirs.double(), code64, w64, c['plane_cfg'], c['flip_z'], c['activation'],
                                 hash=hash_, density_only=density_only)
        loss = (so * g_sig.double()).sum()
        return loss if density_only else loss + (ro * g_rgb.double()).sum()

    w64 = {k: v.clone().requires_grad_(True) for k, v in weights(tag, torch.float64).items()}
    code64 = code[0].double().clone().requires_grad_(True)
    table = G[f'{tag}_table'] if c['ingp'] else None
    oracle_loss(w64, code64, table).backward()

    for p in eng.parameters().values():
        p.requires_grad_(True)
    code_d = code.cuda().requires_grad_(True)
    sig, rgb, n = eng.point_decode_autograd([xyz.cuda()], [dirs.cuda()], code_d, density_only=density_only)
    assert n == [N] and (rgb is None) == density_only
    loss = (sig * g_sig.cuda()).sum()
    if not density_only:
        loss = loss + (rgb * g_rgb.cuda()).sum()
    loss.backward()

    def close(got, want, what):
        want = want.float()
        scale = want.abs().max().item()
        err = (got.cpu() - want).abs().max().item()
        assert err <= 2e-4 * scale + 1e-6, f'{tag} {what}: max |d| {err:.3e} against a gradient of magnitude {scale:.3e}'

    close(code_d.grad[0], code64.grad, 'code planes')
    colour = ('color_net.0.weight', 'color_net.0.bias', 'color_net.2.weight', 'color_net.2.bias')
    for name, p in eng.parameters().items():
        if name == 'encoder.params':
            continue
        if density_only and name in colour:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
            continue
        close(p.grad.reshape(w64[_ONAME[name]].shape), w64[_ONAME[name]].grad, name)
    if c['ingp']:
        gt = eng.parameters()['encoder.params'].grad.cpu().reshape(-1, 2)
        rng = np.random.default_rng(3)
        D = rng.standard_normal(table.shape).astype(np.float32)
        eps = 1e-2
        w0 = {k: v.detach() for k, v in w64.items()}
        with torch.no_grad():
            fd = (oracle_loss(w0, code64.detach(), table + eps * D) - oracle_loss(w0, code64.detach(), table - eps * D)).item() / (2 * eps)
        an = float((gt.double() * torch.from_numpy(D).double()).sum())
        assert abs(fd - an) <= 2e-3 * max(abs(fd), abs(an)) + 1e-4, (fd, an)
    # repeated backward passes accumulate like any autograd leaf; the forward of point_decode_autograd equals point_decode
    with torch.no_grad():
        s2, r2, _ = eng.point_decode([xyz.cuda()], None if density_only else [dirs.cuda()], code.cuda(), density_only=density_only)
    assert torch.equal(s2, sig.detach()) and (density_only or torch.equal(r2, rgb.detach()))
