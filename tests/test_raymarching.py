"""Ray-marching parity.

not gpu : oracle (oracle/raymarching_oracle.c) against the committed golden vectors and against
          closed-form / property checks.
gpu     : HIP kernels (through the C ABI, via the lib.ops.raymarching mirror) against the oracle on
          identical seeded inputs -- bit-exact for every integer buffer and for marched samples,
          1e-5 relative for composited floats (fast exp) -- plus size-independent properties at
          BASELINE sizes (6 x 512^2 rays).
"""
import os

import numpy as np
import pytest
import torch

from oracle import raymarching as O
from scene import camera_rays, sphere_density_grid, morton_np

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'raymarching_small.npz')
AABB = np.array([-1, -1, -1, 1, 1, 1], np.float32)


def small_case(H=32, S=24, views=3, contract=False, dt_gamma=0.0, bound=1.0, C=1, seed=0, noise=True):
    grid = sphere_density_grid(H, C=C, bound=bound, radius=0.55, seed=seed)
    bits = O.packbits(grid, 0.5)
    o, d = camera_rays(views, S, seed=seed)
    aabb = AABB * bound
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    rng = np.random.default_rng(seed + 1)
    noises = rng.random(o.shape[0], dtype=np.float32) if noise else np.zeros(o.shape[0], np.float32)
    return dict(grid=grid, bits=bits, o=o, d=d, aabb=aabb, nears=nears, fars=fars, noises=noises, H=H, C=C,
                bound=bound, contract=contract, dt_gamma=dt_gamma)


# ----------------------------------------------------------------------------------------------
# CPU: oracle vs golden + properties
# ----------------------------------------------------------------------------------------------
def test_oracle_morton_roundtrip_and_numpy_twin():
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, (4096, 3)).astype(np.int32)
    idx = O.morton3D(c)
    assert (O.morton3D_invert(idx) == c).all()
    assert (idx.view(np.uint32) == morton_np(c[:, 0], c[:, 1], c[:, 2])).all()
    # known answers: (1,0,0)->1, (0,1,0)->2, (0,0,1)->4, (3,3,3)->63, (1023,1023,1023)->2^30-1
    ka = O.morton3D(np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 3, 3], [1023, 1023, 1023]], np.int32))
    assert ka.tolist() == [1, 2, 4, 63, (1 << 30) - 1]


def test_oracle_packbits_known_answer():
    g = np.zeros(16, np.float32)
    g[[0, 3, 9]] = 1.0
    g[15] = 0.5
    assert O.packbits(g, 0.5).tolist() == [0b00001001, 0b10000010]


def test_oracle_near_far_closed_form():
    o = np.array([[0, 0, -3], [0, 0, -3], [5, 5, 5], [0, 0, -0.5]], np.float32)
    d = np.array([[0, 0, 1], [0, 1, 0], [0, 0, 1], [0, 0, 1]], np.float32)
    n, f = O.near_far_from_aabb(o, d, AABB, 0.2)
    assert n[0] == 2 and f[0] == 4
    assert n[1] == np.finfo(np.float32).max and f[1] == np.finfo(np.float32).max     # parallel miss
    assert n[2] == np.finfo(np.float32).max
    assert n[3] == np.float32(0.2) and f[3] == 1.5                                    # inside: clamped to min_near


def test_oracle_march_properties():
    cs = small_case()
    xyzs, dirs, ts, rays = O.march_rays_train(cs['o'], cs['d'], 1.0, cs['bits'], 1, cs['H'], cs['nears'], cs['fars'],
                                              cs['noises'], 0.0, 256)
    off, cnt = rays[:, 0], rays[:, 1]
    assert (off == np.concatenate([[0], np.cumsum(cnt)[:-1]])).all()
    assert cnt.sum() == xyzs.shape[0] and cnt.max() <= 256
    # every sample lies in an occupied cell of the bitfield
    H = cs['H']
    n = np.clip((0.5 * (xyzs + 1) * H), 0, H - 1).astype(np.int64)
    cell = morton_np(n[:, 0], n[:, 1], n[:, 2]).astype(np.int64)
    assert ((cs['bits'][cell // 8] >> (cell % 8)) & 1).all()
    # t strictly increasing along each ray, dt within [dt_min, dt_max]
    for r in np.nonzero(cnt > 1)[0][:200]:
        t = ts[off[r]:off[r] + cnt[r], 0]
        assert (np.diff(t) > 0).all()
    ray_of = O.flatten_rays(rays, xyzs.shape[0])
    assert (dirs == cs['d'][ray_of]).all()


def test_oracle_composite_matches_numpy_and_backward_matches_fd():
    cs = small_case(S=12)
    xyzs, dirs, ts, rays = O.march_rays_train(cs['o'], cs['d'], 1.0, cs['bits'], 1, cs['H'], cs['nears'], cs['fars'],
                                              cs['noises'], 0.0, 128)
    rng = np.random.default_rng(3)
    M = xyzs.shape[0]
    sig = (rng.random(M) * 8).astype(np.float32)
    rgb = rng.random((M, 3)).astype(np.float32)
    w, ws, dep, img = O.composite_rays_train(sig, rgb, ts, rays, 1e-4)
    # independent float64 numpy evaluation of the rendering equation
    for r in np.nonzero(rays[:, 1] > 0)[0][:100]:
        sl = slice(rays[r, 0], rays[r, 0] + rays[r, 1])
        a = 1 - np.exp(-sig[sl].astype(np.float64) * ts[sl, 1])
        T = np.concatenate([[1.0], np.cumprod(1 - a)[:-1]])
        stop = np.nonzero(np.cumprod(1 - a) < 1e-4)[0]
        k = (stop[0] + 1) if len(stop) else len(a)
        wr = (a * T)[:k]
        np.testing.assert_allclose(ws[r], wr.sum(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(img[r], (wr[:, None] * rgb[sl][:k]).sum(0), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(dep[r], (wr / ts[sl, 0][:k]).sum(), rtol=2e-5, atol=1e-6)
    # analytic backward vs central differences of the oracle forward on a few samples
    gw = np.zeros(M, np.float32)
    gws = rng.random(rays.shape[0]).astype(np.float32)
    gd = rng.random(rays.shape[0]).astype(np.float32)
    gi = rng.random((rays.shape[0], 3)).astype(np.float32)
    gs, gc = O.composite_rays_train_backward(gw, gws, gd, gi, sig, rgb, ts, rays, ws, dep, img, 0.0)

    def loss(s, c):
        _, a, b, i = O.composite_rays_train(s, c, ts, rays, 0.0)
        return (a.astype(np.float64) * gws).sum() + (b.astype(np.float64) * gd).sum() + (i.astype(np.float64) * gi).sum()
    for m in rng.integers(0, M, 12):
        e = 1e-2
        sp, sm = sig.copy(), sig.copy()
        sp[m] += e
        sm[m] -= e
        fd = (loss(sp, rgb) - loss(sm, rgb)) / (2 * e)
        assert abs(fd - gs[m]) <= 2e-2 * max(1e-2, abs(fd)), (m, fd, gs[m])


def test_oracle_inference_loop_equals_training_march():
    """The k-step inference marcher, iterated, visits exactly the training marcher's samples."""
    cs = small_case(S=10, noise=False)
    N = cs['o'].shape[0]
    xyzs, dirs, ts, rays = O.march_rays_train(cs['o'], cs['d'], 1.0, cs['bits'], 1, cs['H'], cs['nears'], cs['fars'],
                                              cs['noises'], 0.0, 1024)
    alive = np.arange(N, dtype=np.int32)[cs['nears'] < cs['fars']]
    rays_t = cs['nears'].copy()
    got = [[] for _ in range(N)]
    wsum, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    for it in range(400):
        if len(alive) == 0:
            break
        n_step = 3
        x, _, t = O.march_rays(len(alive), n_step, alive, rays_t, cs['o'], cs['d'], 1.0, cs['bits'], 1, cs['H'],
                               cs['nears'], cs['fars'], np.zeros(len(alive), np.float32), 0.0, 1024)
        x = x.reshape(len(alive), n_step, 3)
        t = t.reshape(len(alive), n_step, 2)
        for k, r in enumerate(alive):
            for s in range(n_step):
                if t[k, s, 0] != 0:
                    got[r].append(x[k, s])
        a2 = alive.copy()
        O.composite_rays(len(alive), n_step, a2, rays_t, np.zeros(len(alive) * n_step, np.float32),
                         np.zeros((len(alive) * n_step, 3), np.float32), t.reshape(-1, 2), wsum, dep, img, 1e-2)
        alive = a2[a2 >= 0]
    assert len(alive) == 0
    for r in range(N):
        exp = xyzs[rays[r, 0]:rays[r, 0] + rays[r, 1]]
        g = np.array(got[r], np.float32).reshape(-1, 3)
        assert g.shape == exp.shape and (g == exp).all()


def _golden_outputs():
    cs = small_case(H=32, S=16, views=2, seed=5)
    out = {}
    out['bits'] = cs['bits']
    out['nears'], out['fars'] = cs['nears'], cs['fars']
    for tag, contract, dtg, C, bound in (('a', False, 0.0, 1, 1.0), ('b', True, 1.0 / 128, 2, 2.0)):
        c2 = small_case(H=32, S=16, views=2, seed=5, C=C, bound=bound)
        xyzs, dirs, ts, rays = O.march_rays_train(c2['o'], c2['d'], bound, c2['bits'], C, 32, c2['nears'], c2['fars'],
                                                  c2['noises'], dtg, 192, contract)
        rng = np.random.default_rng(9)
        sig = (rng.random(xyzs.shape[0]) * 6).astype(np.float32)
        rgb = rng.random((xyzs.shape[0], 3)).astype(np.float32)
        w, ws, dep, img = O.composite_rays_train(sig, rgb, ts, rays, 1e-4)
        out.update({f'xyzs_{tag}': xyzs, f'ts_{tag}': ts, f'rays_{tag}': rays, f'w_{tag}': w, f'ws_{tag}': ws,
                    f'dep_{tag}': dep, f'img_{tag}': img})
    return out


def test_oracle_matches_committed_golden():
    """tests/golden/raymarching_small.npz was produced by tests/golden/make_raymarching_golden.py."""
    assert os.path.exists(GOLDEN), 'golden fixture missing'
    gold = np.load(GOLDEN)
    out = _golden_outputs()
    for k in gold.files:
        if out[k].dtype.kind in 'iu':
            assert (out[k] == gold[k]).all(), k
        elif k.startswith(('xyzs', 'ts', 'nears', 'fars')):
            assert (out[k] == gold[k]).all(), k   # marched samples: bit exact
        else:
            np.testing.assert_allclose(out[k], gold[k], rtol=1e-6, atol=1e-7, err_msg=k)


# ----------------------------------------------------------------------------------------------
# GPU: HIP vs oracle
# ----------------------------------------------------------------------------------------------
def _t(a, dev='cuda'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.gpu
def test_gpu_utils_bit_exact(lib):
    from mvedit_amd import raymarching as G
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, (100003, 3)).astype(np.int32)
    idx = G.morton3D(_t(c))
    assert (idx.cpu().numpy() == O.morton3D(c)).all()
    assert (G.morton3D_invert(idx).cpu().numpy() == c).all()
    assert G.morton3D(torch.zeros(0, 3, dtype=torch.int32, device='cuda')).numel() == 0      # empty input
    for n_cells in (8, 24, 8 * 1027, 64 ** 3):
        g = rng.random(n_cells).astype(np.float32)
        g[::7] = 0.5  # ties sit exactly on the threshold
        bits = G.packbits(_t(g).view(1, -1), 0.5)
        assert (bits.cpu().numpy() == O.packbits(g, 0.5)).all(), n_cells
    o, d = camera_rays(3, 40, seed=2)
    d[5] = [0, 1, 0]     # axis-parallel directions -> inf reciprocals
    o[7] = [0, 0, -0.5]
    d[7] = [0, 0, 1]
    n_h, f_h = G.near_far_from_aabb(_t(o), _t(d), _t(AABB), 0.2)
    n_o, f_o = O.near_far_from_aabb(o, d, AABB, 0.2)
    assert (n_h.cpu().numpy() == n_o).all() and (f_h.cpu().numpy() == f_o).all()


@pytest.mark.gpu
@pytest.mark.parametrize('contract,dt_gamma,C,bound', [(False, 0.0, 1, 1.0), (False, 1.0 / 128, 1, 1.0),
                                                       (True, 1.0 / 128, 2, 2.0), (False, 0.0, 3, 4.0)])
def test_gpu_march_train_bit_exact(lib, contract, dt_gamma, C, bound):
    from mvedit_amd import raymarching as G
    cs = small_case(H=64, S=48, views=3, contract=contract, dt_gamma=dt_gamma, bound=bound, C=C)
    xo, do_, to, ro = O.march_rays_train(cs['o'], cs['d'], bound, cs['bits'], C, 64, cs['nears'], cs['fars'],
                                         cs['noises'], dt_gamma, 512, contract)
    xh, dh, th, rh = G.march_rays_train(_t(cs['o']), _t(cs['d']), bound, _t(cs['bits']), C, 64, _t(cs['nears']),
                                        _t(cs['fars']), True, dt_gamma, 512, contract, noises=_t(cs['noises']))
    assert (rh.cpu().numpy() == ro).all(), 'ray (offset,count) table differs'
    assert xh.shape[0] == xo.shape[0] > 1000
    assert (xh.cpu().numpy() == xo).all() and (th.cpu().numpy() == to).all() and (dh.cpu().numpy() == do_).all()
    ray_of = G.flatten_rays(rh, xh.shape[0]).cpu().numpy()
    assert (ray_of == O.flatten_rays(ro, xo.shape[0])).all()


@pytest.mark.gpu
def test_gpu_composite_train_fwd_bwd(lib):
    from mvedit_amd import raymarching as G
    cs = small_case(H=64, S=48, views=3)
    xo, do_, to, ro = O.march_rays_train(cs['o'], cs['d'], 1.0, cs['bits'], 1, 64, cs['nears'], cs['fars'],
                                         cs['noises'], 0.0, 512)
    rng = np.random.default_rng(4)
    M, N = xo.shape[0], ro.shape[0]
    sig = (rng.random(M) * 8).astype(np.float32)
    rgb = rng.random((M, 3)).astype(np.float32)
    ro2 = ro.copy()
    ro2[3] = (M - 2, 7)          # a ray that overruns M -> zero outputs (raymarching.cu:526-533)
    for binarize in (False, True):
        w_o, ws_o, d_o, i_o = O.composite_rays_train(sig, rgb, to, ro2, 1e-4, binarize)
        s_t, c_t = _t(sig).requires_grad_(True), _t(rgb).requires_grad_(True)
        w_h, ws_h, d_h, i_h = G.composite_rays_train(s_t, c_t, _t(to), _t(ro2), 1e-4, binarize)
        for a, b in ((w_h, w_o), (ws_h, ws_o), (d_h, d_o), (i_h, i_o)):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b, rtol=1e-5, atol=1e-6)
        gw = rng.random(M).astype(np.float32)
        gws, gd = rng.random(N).astype(np.float32), rng.random(N).astype(np.float32)
        gi = rng.random((N, 3)).astype(np.float32)
        (w_h * _t(gw)).sum().add((ws_h * _t(gws)).sum()).add((d_h * _t(gd)).sum()).add((i_h * _t(gi)).sum()).backward()
        gs_o, gc_o = O.composite_rays_train_backward(gw, gws, gd, gi, sig, rgb, to, ro2, ws_o, d_o, i_o, 1e-4, binarize)
        np.testing.assert_allclose(s_t.grad.cpu().numpy(), gs_o, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(c_t.grad.cpu().numpy(), gc_o, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_gpu_inference_loop_matches_oracle_loop(lib):
    """Drive the reference's eval loop (base_volume_renderer.py:290-323) on both sides."""
    from mvedit_amd import raymarching as G
    cs = small_case(H=64, S=40, views=2, noise=False)
    N = cs['o'].shape[0]
    rng = np.random.default_rng(8)

    def decode(x):   # stand-in radiance field, evaluated in float32 numpy on both sides
        s = (20 * np.exp(-8 * (x * x).sum(-1))).astype(np.float32)
        c = (0.5 + 0.5 * np.sin(7 * x)).astype(np.float32)
        return s, c

    o_t, d_t, b_t = _t(cs['o']), _t(cs['d']), _t(cs['bits'])
    n_t, f_t = _t(cs['nears']), _t(cs['fars'])
    alive_o = np.arange(N, dtype=np.int32)
    alive_h = torch.arange(N, dtype=torch.int32, device='cuda')
    rt_o = cs['nears'].copy()
    rt_h = n_t.clone()
    acc_o = [np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)]
    acc_h = [torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda'), torch.zeros(N, 3, device='cuda')]
    step = 0
    while step < 256 and len(alive_o) > 0:
        n_alive = len(alive_o)
        n_step = max(min(N // n_alive, 8), 1)
        xo, _, to = O.march_rays(n_alive, n_step, alive_o, rt_o, cs['o'], cs['d'], 1.0, cs['bits'], 1, 64, cs['nears'],
                                 cs['fars'], np.zeros(n_alive, np.float32), 0.0, 256)
        xh, _, th = G.march_rays(n_alive, n_step, alive_h, rt_h, o_t, d_t, 1.0, b_t, 1, 64, n_t, f_t, False, 0.0, 256)
        assert (xh.cpu().numpy() == xo).all() and (th.cpu().numpy() == to).all(), f'march differs at round {step}'
        s, c = decode(xo)
        O.composite_rays(n_alive, n_step, alive_o, rt_o, s, c, to, *acc_o, 1e-2)
        G.composite_rays(n_alive, n_step, alive_h, rt_h, _t(s), _t(c), th, *acc_h, 1e-2)
        assert (alive_h.cpu().numpy() == alive_o).all()
        comp, n_kept = G.compact_alive(alive_h, n_alive)
        alive_o = alive_o[alive_o >= 0]
        assert int(n_kept.item()) == len(alive_o)
        alive_h = comp[:len(alive_o)].contiguous()
        assert (alive_h.cpu().numpy() == alive_o).all()
        step += n_step
    for a, b in zip(acc_h, acc_o):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(rt_h.cpu().numpy(), rt_o)


@pytest.mark.gpu
def test_gpu_full_size_properties(lib):
    """BASELINE size (render_bs=6 views x 512^2 rays, 128^3 grid): size-independent properties only."""
    from mvedit_amd import raymarching as G
    H = 128
    grid = sphere_density_grid(H, radius=0.5)
    bits = G.packbits(_t(grid), 0.5)
    o, d = camera_rays(6, 512, seed=1, jitter=False)
    o_t, d_t = _t(o), _t(d)
    nears, fars = G.near_far_from_aabb(o_t, d_t, _t(AABB), 0.2)
    xyzs, dirs, ts, rays = G.march_rays_train(o_t, d_t, 1.0, bits, 1, H, nears, fars, False, 0.0, 1024)
    cnt, off = rays[:, 1].long(), rays[:, 0].long()
    M = xyzs.shape[0]
    assert int(cnt.sum()) == M and M > 10 ** 6
    assert torch.equal(off, torch.cumsum(cnt, 0) - cnt)                 # offsets = exclusive scan in ray order
    n = (0.5 * (xyzs.double() + 1) * H).clamp(0, H - 1).long()
    cell = torch.from_numpy(morton_np(*(n[:, i].cpu().numpy() for i in range(3))).astype(np.int64)).cuda()
    assert bool(((bits[cell // 8].long() >> (cell % 8)) & 1).all())      # every sample is in an occupied cell
    assert bool((xyzs.norm(dim=-1) < 0.5 + 2 * 1.74 / H).all())          # ...which lies within the sphere
    ray_of = G.flatten_rays(rays, M).long()
    assert torch.equal(dirs, d_t[ray_of])
    # a sub-sample of rays, exact against the oracle
    pick = np.random.default_rng(0).choice(o.shape[0], 20000, replace=False)
    pick.sort()
    xo, _, to, ro = O.march_rays_train(o[pick], d[pick], 1.0, bits.cpu().numpy(), 1, H, nears.cpu().numpy()[pick],
                                       fars.cpu().numpy()[pick], np.zeros(len(pick), np.float32), 0.0, 1024)
    assert (ro[:, 1] == rays[:, 1].cpu().numpy()[pick]).all()
    rr = rays.cpu().numpy()[pick]
    sel = np.concatenate([np.arange(a, a + b) for a, b in rr if b > 0])
    assert (xyzs.cpu().numpy()[sel] == xo).all() and (ts.cpu().numpy()[sel] == to).all()
    # compositing: opaque medium -> alpha saturates to 1 inside the silhouette, 0 outside
    sig = torch.full((M,), 400.0, device='cuda')     # dt = dt_min = 2*sqrt(3)/1024 here: alpha = 0.74 per sample
    rgb = torch.rand(M, 3, device='cuda')
    w, ws, dep, img = G.composite_rays_train(sig, rgb, ts, rays, 1e-4)
    hit = cnt > 16
    assert bool((ws[hit] > 0.99).all()) and bool((ws[cnt == 0] == 0).all())
    assert bool((img.amax(-1) <= ws + 1e-5).all())


def test_oracle_matches_reference_kernel_outputs():
    """tests/golden/raymarching_ref_gfx950.npz holds outputs of the REFERENCE's own kernels
    (oracle/_ref, built from /root/reference by oracle/build_ref.py) run on an MI355X by
    tests/test_raymarching_ref.py (MVE_DUMP_REF_GOLDEN): per-ray sample counts and, per ray in ray
    order, the marched samples.  The C oracle must reproduce them bit-for-bit."""
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'raymarching_ref_gfx950.npz'))
    cs = small_case(H=64, S=48, views=3, contract=False, dt_gamma=0.0, bound=1.0, C=1)
    xo, _, to, ro = O.march_rays_train(cs['o'], cs['d'], 1.0, cs['bits'], 1, 64, cs['nears'], cs['fars'], cs['noises'],
                                       0.0, 512, False)
    assert (ro[:, 1] == gold['rays_counts']).all()
    assert xo.shape[0] == int(gold['n_samples'])
    n = gold['xyzs_head'].shape[0]
    assert (xo[:n] == gold['xyzs_head']).all() and (to[:n] == gold['ts_head']).all()
    np.testing.assert_array_equal(xo.astype(np.float64).sum(0), gold['xyz_sum'])
    np.testing.assert_array_equal(to.astype(np.float64).sum(0), gold['ts_sum'])
