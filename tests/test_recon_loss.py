"""Image-space loss of one NeRF optimisation iteration (csrc/recon_loss.hip + recon_loss_core.h behind mvedit_amd.recon_loss) vs the
reference's OWN statements (lib/pipelines/mvedit_3d_pipeline.py:542-603) executed on the CPU in float64 with torch autograd
(tests/golden/recon_loss_ref.npz, written by tests/golden/make_recon_loss_golden.py).

Bars.  The torch restatement (oracle/recon_loss_oracle.py) equals the reference's values and gradients to 1e-10.  The product arithmetic is
fp32 like the reference's real run; depth -> normal differentiates points at distance 1 / depth, so single precision itself is up to
1e-3 of the gradient scale away from the float64 values at isolated pixels (alpha = 0 next to foreground).  The bar is therefore
relative to what single precision costs the reference: every element must be within 4 x the fp32-torch error at that element plus 1e-5 of
the tensor's scale.  Pixels within two steps of an alpha = 0 pixel (where the differenced points sit at distance 1e6 and the normal is
decided by rounding, in the reference's precision too) are compared at 5 % of the tensor's scale only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import recon_loss_oracle as R

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'recon_loss_ref.npz'))
NC = int(G['n_cases'])
LEAVES = ('image', 'weights_sum', 'depth', 'weights')
PARTS = ('loss', 'pixel_rgb_loss', 'alphas_loss', 'normal_reg_loss', 'depth_loss', 'entropy_loss')


def case(i, dtype=torch.float64):
    c = lambda k: G[f'c{i}_{k}']
    t = lambda k: torch.from_numpy(c(k)).to(dtype)
    tm = bool(c('tonemap'))
    args = [t(k) for k in LEAVES] + [t('ts'), t('target_rgbs'), t('target_m_blur'), t('target_dir'), t('cam_w') / float(c('cam_weights_mean')), t('lights')]
    kw = dict(target_n=t('target_n') if c('use_normal') else None, target_depth=t('target_depth') if c('use_depth') else None,
              shaded=(not bool(c('is_init'))) or bool(c('init_shaded')), is_init=bool(c('is_init')), ambient_light=float(c('ambient_light')),
              normal_reg_weight=float(c('normal_reg_weight')), depth_weight=float(c('depth_weight')), entropy_weight=float(c('entropy_weight')),
              bg_width=float(c('bg_width')))
    luts = (torch.from_numpy(G['lut_x']).to(dtype), torch.from_numpy(G['lut_y']).to(dtype)) if tm else (None, None)
    return args, kw, luts, c


def oracle_run(args, kw, luts, ext=None, gl=1.0):
    """-> (res, grads) of gl * loss + <ext_rgb, out_rgbs> + <ext_nrm, out_normals> by torch autograd over the restatement"""
    leaves = [a.clone().requires_grad_(True) for a in args[:4]]
    res = R.nerf_optim_loss(*leaves, args[4][:, 1], *args[5:], lut_x=luts[0], lut_y=luts[1], **kw)
    total = res['loss'] * gl
    if ext is not None:
        total = total + (res['out_rgbs'] * ext[0]).sum() + (res['out_normals'] * ext[1]).sum()
    return res, torch.autograd.grad(total, leaves)


def well_conditioned(alpha, P, ps):
    """[P * ps * ps] bool: no alpha = 0 pixel within two steps (the support of a pixel's normal and of the gradients through it)"""
    z = torch.as_tensor(np.asarray(alpha)).reshape(P, 1, ps, ps) <= 0
    near = torch.nn.functional.max_pool2d(z.float(), 5, stride=1, padding=2) > 0
    return ~near.reshape(-1).numpy()


def close_enough(got, gold, fp32_ref, what, ok=None):
    got, gold, fp32_ref = (np.asarray(a, np.float64).reshape(-1) for a in (got, gold, fp32_ref))
    scale = max(np.abs(gold).max(), 1e-12)
    bar = 4 * np.abs(fp32_ref - gold) + 1e-5 * scale
    if ok is not None:
        bar = np.where(np.repeat(ok, got.size // ok.size), bar, 0.05 * scale)
    err = np.abs(got - gold)
    j = int(np.argmax(err - bar))
    assert err[j] <= bar[j], f'{what}: |err| {err[j]:.3e} > bar {bar[j]:.3e} at {j} (got {got[j]:.6e}, reference {gold[j]:.6e})'


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize('i', range(NC))
def test_oracle_restatement_equals_reference_statements(i):
    args, kw, luts, c = case(i)
    res, grads = oracle_run(args, kw, luts)
    for k in PARTS:
        if k == 'depth_loss' and not c('use_depth'):
            continue
        assert abs(float(res[k].detach()) - float(c(k))) <= 1e-10 * max(1.0, abs(float(c(k)))), k
    for k in ('out_rgbs', 'out_normals', 'out_normals_fg'):
        assert np.abs(res[k].detach().numpy() - c(k)).max() <= 1e-12, k
    assert np.array_equal(res['out_normals_fg_weight'].squeeze(1).numpy(), c('out_normals_fg_weight').squeeze(-1))
    for g, k in zip(grads, LEAVES):
        assert np.abs(g.numpy() - c('g_' + k)).max() <= 1e-10 * max(1.0, np.abs(c('g_' + k)).max()), k


def _host(args, kw, luts, **extra):
    from oracle import devcore as D
    a = [x.numpy() for x in args]
    kw = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in kw.items()}
    return D.recon_loss(*a, lut_x=None if luts[0] is None else luts[0].numpy(), lut_y=None if luts[0] is None else luts[1].numpy(), **kw, **extra)


@pytest.mark.parametrize('i', range(NC))
def test_kernel_arithmetic_host_build_vs_reference(i):
    """recon_loss_core.h -- the source the HIP kernels are made of -- built for the host and run in the kernels' launch order."""
    args, kw, luts, c = case(i)
    a32, kw32, luts32, _ = case(i, torch.float32)
    res32, g32 = oracle_run(a32, kw32, luts32)
    h = _host(args, kw, luts)
    ok = well_conditioned(c('weights_sum'), int(c('P')), int(c('ps')))
    gold = [float(c(k)) for k in PARTS]
    assert np.abs(h['losses'] - np.asarray(gold)).max() <= 2e-5 * max(1.0, np.abs(gold).max())
    for k in ('out_rgbs', 'out_normals'):
        close_enough(h[k], c(k), res32[k].detach().numpy(), k, ok)
    for k, g in zip(LEAVES, g32):
        close_enough(h['g_' + k], c('g_' + k), g.numpy(), 'g_' + k, None if k == 'weights' else ok)


@pytest.mark.parametrize('i', (0, 1))
def test_host_build_external_gradients_and_scale(i):
    """the patch losses of :611-627 reach the kernels as gradients w.r.t. out_rgbs / out_normals; autograd's incoming scale as g_loss"""
    args, kw, luts, c = case(i)
    g = torch.Generator().manual_seed(5 + i)
    ext = [torch.randn(*c('out_rgbs').shape, generator=g, dtype=torch.float64) * 1e-3 for _ in range(2)]
    _, gold = oracle_run(args, kw, luts, ext, gl=0.37)
    a32, kw32, luts32, _ = case(i, torch.float32)
    _, g32 = oracle_run(a32, kw32, luts32, [e.float() for e in ext], gl=0.37)
    h = _host(args, kw, luts, g_rgb_ext=ext[0].numpy(), g_nrm_ext=ext[1].numpy(), gl=0.37)
    ok = well_conditioned(c('weights_sum'), int(c('P')), int(c('ps')))
    for k, gd, gs in zip(LEAVES, gold, g32):
        close_enough(h['g_' + k], gd.numpy(), gs.numpy(), 'g_' + k, None if k == 'weights' else ok)


def test_descriptor_layout_matches_the_header(tmp_path):
    """ctypes mirror of MveReconLossDesc vs the C compiler's layout of include/mvedit_amd.h"""
    pytest.importorskip('mvedit_amd._lib')
    from mvedit_amd.recon_loss import _Desc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f[0] for f in _Desc._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mvedit_amd.h"\nint main(void) {\n  printf("%zu", sizeof(MveReconLossDesc));\n'
                   + ''.join(f'  printf(" %zu", offsetof(MveReconLossDesc, {f}));\n' for f in fields) + '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)], check=True)
    nums = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(_Desc)
    assert nums[1:] == [getattr(_Desc, f).offset for f in fields]


# ------------------------------------------------------------------------------------------------ GPU
def _gpu_run(args, kw, luts, ext=None, gl=None):
    from mvedit_amd.recon_loss import nerf_optim_loss
    from mvedit_amd.tonemapping import Tonemapping
    tm = None
    if luts[0] is not None:
        tm = Tonemapping(device='cuda')
        assert np.array_equal(tm.lut_x.cpu().numpy(), luts[0].float().numpy())
    cu = lambda v: None if v is None else v.float().cuda()
    leaves = [cu(a).requires_grad_(True) for a in args[:4]]
    kw = {k: (cu(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    res = nerf_optim_loss(*leaves, *[cu(a) for a in args[4:]], tonemapping=tm, **kw)
    total = res['loss'] * (1.0 if gl is None else gl)
    if ext is not None:
        total = total + (res['out_rgbs'] * cu(ext[0])).sum() + (res['out_normals'] * cu(ext[1])).sum()
    grads = torch.autograd.grad(total, leaves)
    return res, [g.cpu().numpy() for g in grads]


@pytest.mark.gpu
@pytest.mark.parametrize('i', range(NC))
def test_hip_vs_reference_and_host_build(lib, i):
    args, kw, luts, c = case(i)
    a32, kw32, luts32, _ = case(i, torch.float32)
    res32, g32 = oracle_run(a32, kw32, luts32)
    res, grads = _gpu_run(args, kw, luts)
    h = _host(args, kw, luts)
    ok = well_conditioned(c('weights_sum'), int(c('P')), int(c('ps')))
    gold = np.asarray([float(c(k)) for k in PARTS])
    got = np.asarray([float(res[k]) for k in PARTS])
    assert np.abs(got - gold).max() <= 2e-5 * max(1.0, np.abs(gold).max()), (got, gold)
    for k in ('out_rgbs', 'out_normals'):
        close_enough(res[k].detach().cpu().numpy(), c(k), res32[k].detach().numpy(), k, ok)
    for k, g, gs in zip(LEAVES, grads, g32):
        okk = None if k == 'weights' else ok
        close_enough(g, c('g_' + k), gs.numpy(), 'g_' + k, okk)
        # device vs host build of the same source: only libm (log, log2, sqrt, division) may differ in the last place
        close_enough(g, h['g_' + k], h['g_' + k], 'device vs host g_' + k, okk)


@pytest.mark.gpu
def test_hip_external_gradients_scale_and_determinism(lib):
    args, kw, luts, c = case(0)
    g = torch.Generator().manual_seed(5)
    ext = [torch.randn(*c('out_rgbs').shape, generator=g, dtype=torch.float64) * 1e-3 for _ in range(2)]
    _, gold = oracle_run(args, kw, luts, ext, gl=0.37)
    a32, kw32, luts32, _ = case(0, torch.float32)
    _, g32 = oracle_run(a32, kw32, luts32, [e.float() for e in ext], gl=0.37)
    res1, grads1 = _gpu_run(args, kw, luts, ext, gl=0.37)
    res2, grads2 = _gpu_run(args, kw, luts, ext, gl=0.37)
    ok = well_conditioned(c('weights_sum'), int(c('P')), int(c('ps')))
    for k, got, gd, gs in zip(LEAVES, grads1, gold, g32):
        close_enough(got, gd.numpy(), gs.numpy(), 'g_' + k, None if k == 'weights' else ok)
    assert all(np.array_equal(a, b) for a, b in zip(grads1, grads2)) and torch.equal(res1['loss'], res2['loss'])      # gathers + fixed-order sums


@pytest.mark.gpu
def test_hip_full_size_iteration(lib):
    """the reference's working size: 8 patches of 128 x 128 rays (n_inverse_rays = 2^17, lib/pipelines/utils.py:233), ~8 samples per ray"""
    P, ps = 8, 128
    g = torch.Generator().manual_seed(21)
    N = P * ps * ps
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, ps), torch.linspace(-1, 1, ps), indexing='ij')
    dirs = torch.stack([xx * 0.27, yy * 0.27, torch.ones_like(xx)], -1)[None].expand(P, -1, -1, -1).contiguous()
    alpha = (0.05 + torch.rand(N, generator=g) * 1.1).clamp(0, 1)
    alpha.view(P, ps, ps)[0, :9, :40] = 0.0
    depth = (0.25 + 0.05 * torch.sin(4 * xx)[None].expand(P, -1, -1).reshape(N) + 0.02 * torch.rand(N, generator=g)) * alpha
    image = torch.rand(N, 3, generator=g) * alpha[:, None]
    M = 8 * N
    weights = torch.rand(M, generator=g) * 0.2
    ts = torch.stack([torch.rand(M, generator=g) * 3 + 1, torch.rand(M, generator=g) * 0.03 + 1e-3], -1)
    args = [image, alpha, depth, weights, ts, torch.rand(P, ps, ps, 3, generator=g), torch.rand(P, ps, ps, 1, generator=g), dirs,
            torch.rand(P, generator=g) + 0.5, torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)]
    kw = dict(target_n=torch.rand(P, ps, ps, 3, generator=g), target_depth=torch.rand(P, ps, ps, 1, generator=g) * 0.4, shaded=True, is_init=False,
              ambient_light=0.2, normal_reg_weight=2.0, depth_weight=0.5, entropy_weight=1.0, bg_width=0.015)
    luts = (torch.from_numpy(G['lut_x']), torch.from_numpy(G['lut_y']))
    res64, g64 = oracle_run([a.double() for a in args], {k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()},
                            [l.double() for l in luts])
    res32, g32 = oracle_run(args, kw, [l.float() for l in luts])
    res, grads = _gpu_run(args, kw, luts)
    for k in PARTS:
        assert abs(float(res[k]) - float(res64[k])) <= 2e-5 * max(1.0, abs(float(res64[k]))), k
    ok = well_conditioned(alpha, P, ps)
    for k, got, gd, gs in zip(LEAVES, grads, g64, g32):
        close_enough(got, gd.numpy(), gs.numpy(), 'g_' + k, None if k == 'weights' else ok)


@pytest.mark.gpu
def test_hip_timing_against_the_torch_statements_on_the_same_gpu(lib):
    """not a parity test: prints what one iteration's image-space loss costs natively and as the reference's torch statements (the
    restatement, on the GPU) -- the number DESIGN.md section 4.8 quotes comes from here"""
    import time
    P, ps = 8, 128
    g = torch.Generator().manual_seed(3)
    N, M = P * ps * ps, 8 * P * ps * ps
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, ps), torch.linspace(-1, 1, ps), indexing='ij')
    dirs = torch.stack([xx * 0.27, yy * 0.27, torch.ones_like(xx)], -1)[None].expand(P, -1, -1, -1).contiguous()
    alpha = (0.05 + torch.rand(N, generator=g) * 1.1).clamp(0, 1)
    args = [torch.rand(N, 3, generator=g) * alpha[:, None], alpha, (0.25 + 0.02 * torch.rand(N, generator=g)) * alpha, torch.rand(M, generator=g) * 0.2,
            torch.stack([torch.rand(M, generator=g) * 3 + 1, torch.rand(M, generator=g) * 0.03 + 1e-3], -1), torch.rand(P, ps, ps, 3, generator=g),
            torch.rand(P, ps, ps, 1, generator=g), dirs, torch.rand(P, generator=g) + 0.5, torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)]
    kw = dict(target_n=torch.rand(P, ps, ps, 3, generator=g), target_depth=None, shaded=True, is_init=False, ambient_light=0.2, normal_reg_weight=2.0,
              entropy_weight=1.0, bg_width=0.015)
    from mvedit_amd.recon_loss import nerf_optim_loss
    from mvedit_amd.tonemapping import Tonemapping
    tm = Tonemapping(device='cuda')
    cu = [a.cuda() for a in args]
    kwc = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}

    def native():
        leaves = [a.clone().requires_grad_(True) for a in cu[:4]]
        nerf_optim_loss(*leaves, *cu[4:], tonemapping=tm, **kwc)['loss'].backward()

    def torch_ops():
        leaves = [a.clone().requires_grad_(True) for a in cu[:4]]
        R.nerf_optim_loss(*leaves, cu[4][:, 1], *cu[5:], lut_x=tm.lut_x, lut_y=tm.lut_y, **kwc)['loss'].backward()
    for name, fn in (('native', native), ('torch statements', torch_ops)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print(f'recon loss fwd+bwd, 8 x 128^2 rays, {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms')


@pytest.mark.gpu
def test_nerf_optim_iteration_on_native_kernels_only(lib):
    """The NeRF half of the reconstruct step wired together the way `nerf_optim` wires it (mvedit_3d_pipeline.py:507-633), every stage
    native: VolumeRenderer training forward (march -> cull -> decode -> composite) -> nerf_optim_loss (shading, tone mapping, L1, TV,
    entropy) -> backward through composite and the hash grid -> Adam on the decoder parameters.  The colour + alpha terms must fall and
    every parameter gradient must be finite."""
    from mvedit_amd.nerf import VolumeRenderer
    from mvedit_amd.recon_loss import nerf_optim_loss
    from mvedit_amd.tonemapping import Tonemapping
    from oracle import raymarching as ORM
    from scene import camera_rays as scene_rays, sphere_density_grid
    from test_nerf import _decoder
    _, dec = _decoder(12, 320, table_scale=0.1)
    H, P, ps = 64, 2, 32
    bits = torch.from_numpy(ORM.packbits(sphere_density_grid(H, radius=0.6), 0.5)).cuda()
    o, d = scene_rays(P, ps, seed=5)                                              # view-major, row-major: one patch per view
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    dec.max_steps = 256
    for t in dec.parameters().values():
        t.requires_grad_(True)
    vr = VolumeRenderer(dec)
    vr.training = True
    fl = ps / (2 * np.tan(np.deg2rad(15)))
    ys, xs = torch.meshgrid(torch.arange(ps, dtype=torch.float32), torch.arange(ps, dtype=torch.float32), indexing='ij')
    dirs = torch.stack([(xs + 0.5 - ps / 2) / fl, (ys + 0.5 - ps / 2) / fl, torch.ones_like(xs)], -1)[None].repeat(P, 1, 1, 1).cuda()
    with torch.no_grad():
        hit = vr.forward(o, d, bits, H)['weights_sum'].reshape(P, ps, ps, 1) > 0.0   # rays that meet the occupied sphere at all
    target_m = hit.float()
    target_rgbs = torch.tensor([0.8, 0.3, 0.2], device='cuda').expand(P, ps, ps, 3) * target_m + (1 - target_m)
    tm = Tonemapping(device='cuda')
    # the patch loss of :611-616 on top: LPIPS (bf16, synthetic VGG weights) over out_rgbs -- its gradient re-enters the native backward
    from mvedit_amd.lpips import LPIPSEngine
    from test_lpips import _case as lpips_case
    lp = LPIPSEngine.from_state_dict({k: v.to(torch.bfloat16).float() for k, v in lpips_case()[0].items()}, torch.bfloat16)
    opt = torch.optim.Adam(list(dec.parameters().values()), lr=1e-2, eps=1e-15)
    hist = []
    for it in range(30):
        opt.zero_grad()
        out = vr.forward(o, d, bits, H, dt_gamma=0.0)
        res = nerf_optim_loss(out['image'], out['weights_sum'], out['depth'], out['weights'], out['ts'][0], target_rgbs, target_m, dirs,
                              torch.ones(P, device='cuda'), torch.nn.functional.normalize(torch.tensor([[0.3, -0.5, -1.0]] * P, device='cuda'), dim=-1),
                              tonemapping=tm, shaded=True, normal_reg_weight=0.5, entropy_weight=0.2)
        patch = lp(res['out_rgbs'].permute(0, 3, 1, 2), target_rgbs.permute(0, 3, 1, 2)).mean()
        (res['loss'] + 0.3 * patch).backward()
        for k, t in dec.parameters().items():
            assert t.grad is not None and torch.isfinite(t.grad).all(), (it, k)
        opt.step()
        hist.append((float(res['pixel_rgb_loss']), float(res['alphas_loss']), float(res['normal_reg_loss']), float(res['entropy_loss'])))
    print('nerf_optim loop (rgb, alpha, tv, entropy): first', hist[0], 'last', hist[-1])
    assert hist[-1][0] + hist[-1][1] < 0.7 * (hist[0][0] + hist[0][1]), (hist[0], hist[-1])
