"""Texture-space ops of the mesh path against outputs of the REFERENCE's own functions
(tests/golden/reference_py.npz, produced by executing lib/ops/edge_dilation.py in make_reference_py_golden.py)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'reference_py.npz')


@pytest.mark.gpu
def test_edge_dilation_matches_reference_output(lib):
    from mvedit_amd.mesh_ops import edge_dilation
    g = np.load(GOLD)
    img, mask = torch.from_numpy(g['dil_img']).cuda(), torch.from_numpy(g['dil_mask']).cuda()
    out = edge_dilation(img, mask, radius=3, iters=7)
    assert torch.equal(out.cpu(), torch.from_numpy(g['dil_out_r3_i7'])), 'bit-exact: the op only copies texels'
    out = edge_dilation(img, mask, radius=1, iters=2)
    assert torch.equal(out.cpu(), torch.from_numpy(g['dil_out_r1_i2']))
    assert edge_dilation(img, mask, radius=0) is img
    # atlas-sized input (BASELINE: 1024^2 albedo): idempotent once the mask is full, valid texels never change
    big = torch.rand(1, 3, 1024, 1024, device='cuda')
    m = (torch.rand(1, 1, 1024, 1024, device='cuda') > 0.6).float()
    d = edge_dilation(big * m, m, 3, 7)
    assert torch.equal((d * m), big * m)
    assert (d.amax(1, keepdim=True) > 0).float().mean() > 0.99


def _clip_positions(v, n_views, S, seed=0):
    """Project with the reference's conventions (base_mesh_renderer.py:222-237) in numpy float32."""
    from oracle import nerf_oracle  # noqa: F401  (shared golden poses)
    g = np.load(GOLD)
    poses = g['poses'][:n_views, :3].astype(np.float32)
    f = S / (2 * np.tan(np.deg2rad(15)))
    intr = np.tile(np.array([[f, f, S / 2, S / 2]], np.float32), (n_views, 1))
    return poses, intr


@pytest.mark.gpu
@pytest.mark.parametrize('S,subdiv', [(64, 2), (256, 4), (512, 5)])
def test_rasterize_bit_exact_vs_oracle(lib, S, subdiv):
    from mvedit_amd.mesh_ops import MeshRenderer, rasterize, interpolate
    from oracle import raster as OR
    from scene import icosphere
    v, f = icosphere(subdiv, 0.6)
    v = v + np.random.default_rng(0).normal(0, 0.004, v.shape).astype(np.float32)     # break symmetry / create thin slivers
    poses, intr = _clip_positions(v, 3, S)
    mr = MeshRenderer(near=0.01, far=100)
    v_cam, v_clip, _ = mr.project(torch.from_numpy(v).cuda(), torch.from_numpy(poses).cuda(), torch.from_numpy(intr).cuda(), S, S)
    rast_h = rasterize(v_clip, torch.from_numpy(f).cuda(), (S, S))
    rast_o = OR.rasterize(v_clip.cpu().numpy(), f, (S, S))
    ids_h, ids_o = rast_h[..., 3].cpu().numpy(), rast_o[..., 3]
    assert (ids_h == ids_o).all(), f'triangle-id buffer differs at {(ids_h != ids_o).sum()} pixels'
    assert (rast_h.cpu().numpy() == rast_o).all(), 'u, v, z/w must be bit-exact too (same float ops, no contraction)'
    cover = (ids_o > 0).mean()
    assert 0.05 < cover < 0.6
    # interpolation of arbitrary attributes (own index buffer), broadcast and per-view
    rng = np.random.default_rng(1)
    attr = rng.normal(size=(1, v.shape[0], 5)).astype(np.float32)
    np.testing.assert_allclose(interpolate(torch.from_numpy(attr).cuda(), rast_h, torch.from_numpy(f).cuda()).cpu().numpy(),
                               OR.interpolate(attr, rast_o, f), rtol=1e-6, atol=1e-6)
    out = interpolate(v_cam[..., 2:3].contiguous(), rast_h, torch.from_numpy(f).cuda()).cpu().numpy()
    np.testing.assert_allclose(out, OR.interpolate(v_cam[..., 2:3].cpu().numpy(), rast_o, f), rtol=1e-6, atol=1e-6)
    # a triangle far larger than the small-box limit (swept by a whole block) in front of everything, plus degenerate input
    big = np.array([[[-0.9, -0.9, -0.5, 1], [0.9, -0.8, -0.5, 1], [0.0, 0.9, -0.5, 1], [0, 0, 0, 0], [0.5, 0.5, 0.2, -1]]], np.float32)
    tri = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 4], [1, 1, 2]], np.int32)
    r_h = rasterize(torch.from_numpy(big).cuda(), torch.from_numpy(tri).cuda(), (200, 300)).cpu().numpy()
    r_o = OR.rasterize(big, tri, (200, 300))
    assert (r_h == r_o).all() and (r_o[..., 3] == 1).sum() > 10000 and (r_o[..., 3] > 1).sum() == 0


@pytest.mark.gpu
def test_mesh_renderer_geometry_seam(lib):
    """Rendering the icosphere with the reference's camera conventions: depth = 1/z of the front surface, outward normals."""
    from mvedit_amd.mesh_ops import MeshRenderer
    from scene import icosphere
    v, f = icosphere(4, 0.6)
    vn = v / np.linalg.norm(v, axis=-1, keepdims=True)
    S = 128
    poses, intr = _clip_positions(v, 4, S)
    mr = MeshRenderer(near=0.01, far=100)
    t = lambda a: torch.from_numpy(a).cuda()
    out = mr.render_geometry(t(v), t(f), t(vn.astype(np.float32)), t(f), t(poses), t(intr), S, S)
    alpha, depth, normal = out['alpha'][..., 0], out['depth'], out['normal']
    # silhouette: a sphere of radius 0.6 seen from 3.7 -> angular radius asin(0.6/3.7); fov 30 deg
    frac = np.pi * (np.tan(np.arcsin(0.6 / 3.7)) / np.tan(np.deg2rad(15))) ** 2 / 4
    assert abs(alpha.mean().item() - frac) < 0.01
    c = S // 2
    assert abs(1 / depth[:, c, c].mean().item() - (3.7 - 0.6)) < 0.02                 # centre pixel sees the nearest point
    # camera-space normal at the centre points at the camera: (0,0,1) -> colour (0.5, 0.5, 1.0)
    np.testing.assert_allclose(normal[:, c, c].mean(0).cpu().numpy(), [0.5, 0.5, 1.0], atol=0.03)
    assert torch.equal(normal[alpha == 0], normal.new_tensor([0.5, 0.5, 1.0]).expand(int((alpha == 0).sum()), 3))


@pytest.mark.gpu
@pytest.mark.parametrize('S,map_size,subdiv,n_views,filt', [(96, 128, 2, 5, 'linear'), (256, 512, 3, 8, 'linear'),
                                                           (64, 128, 2, 5, 'linear-mipmap-linear'), (256, 1024, 3, 6, 'linear-mipmap-linear')])
def test_bake_multiview_vs_oracle(lib, S, map_size, subdiv, n_views, filt):
    """Texture back-projection: every stage against the oracle restatement of base_mesh_renderer.py:507-603 on the SAME
    projected vertices (so both rasterisers see identical input and the id buffers are bit-identical).
    Tolerances: visibility is fixed point 2^-32 per add; the view weight is cos^8 of a normal built from differences of
    nearby points, which amplifies float32 rounding (sqrtf/div order) 8x and more on grazing pixels -> 1e-3 relative."""
    from mvedit_amd.mesh_ops import MeshRenderer, Mesh
    from oracle import bake_oracle as BO
    from scene import icosphere, face_atlas
    v, f = icosphere(subdiv, 0.6)
    v = (v * (1 + 0.15 * np.sin(5 * v[:, :1]))).astype(np.float32)
    vt, ft = face_atlas(f)
    poses, intr = _clip_positions(v, n_views, S)
    rng = np.random.default_rng(4)
    yy, xx = np.meshgrid(np.linspace(0, 1, S, dtype=np.float32), np.linspace(0, 1, S, dtype=np.float32), indexing='ij')
    images = np.stack([np.stack([0.5 + 0.5 * np.sin(7 * xx + i), yy, 0.5 + 0.5 * np.cos(9 * yy * xx + i)], -1) for i in range(n_views)])
    images = (images + rng.normal(0, 0.02, images.shape)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    mr = MeshRenderer(near=0.01, far=100, texture_filter=filt)       # the reference's default is the mip-mapped filter (:196)
    v_cam, v_clip, _ = mr.project(t(v), t(poses), t(intr), S, S)
    # alpha = the mesh's own silhouette, as in the pipeline (the images being baked were rendered from this mesh)
    from oracle import raster as OR
    rast_o = OR.rasterize(v_clip.cpu().numpy(), f, (S, S))
    alphas = (rast_o[..., 3:4] > 0).astype(np.float32)
    mesh = Mesh(t(v), t(f), t(vt), t(ft))
    (mesh,), dbg = mr.bake_multiview([mesh], t(images)[None], t(alphas)[None], t(poses)[None], t(intr)[None], map_size=map_size,
                                     cos_weight_pow=8.0, render_bs=3, return_debug=True)
    # render_bs is a memory knob: chunks of 3 views (the reference's way of walking them) give bitwise the same atlas as one chunk
    mr.min_render_bs = 1
    mesh3 = Mesh(t(v), t(f), t(vt), t(ft))
    (mesh3,), dbg3 = mr.bake_multiview([mesh3], t(images)[None], t(alphas)[None], t(poses)[None], t(intr)[None], map_size=map_size,
                                       cos_weight_pow=8.0, render_bs=3, return_debug=True)
    assert len(dbg3['vis']) == -(-n_views // 3) and len(dbg['vis']) == 1
    assert torch.equal(dbg3['accum'], dbg['accum']) and torch.equal(mesh3.albedo, mesh.albedo)
    alb_o, accum_o, valid_o, dbg_o = BO.bake_multiview(v, f, vt, ft, images, alphas, poses, intr, map_size, 8.0,
                                                      projected=(v_cam.cpu().numpy(), v_clip.cpu().numpy()), texture_filter=filt)
    assert (dbg['tex_rast'].cpu().numpy() == dbg_o['tex_rast']).all(), 'UV-space raster must be bit-exact'
    assert (dbg['valid'].cpu().numpy() == valid_o).all() and 0.2 < valid_o.mean() < 0.6
    vis_h = torch.cat(dbg['vis']).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(vis_h, dbg_o['vis'], rtol=0, atol=2e-6 if filt == 'linear' else 1e-5)
    assert dbg_o['vis'].max() > (1.0 if filt == 'linear' else 0.05) and (dbg_o['vis'] > 0).mean() > 0.02
    np.testing.assert_allclose(torch.cat(dbg['wimg']).cpu().numpy(), dbg_o['wimg'], rtol=1e-3, atol=2e-6)
    assert dbg_o['wimg'].max() > 0.5
    acc_h = dbg['accum'].cpu().numpy()
    np.testing.assert_allclose(acc_h, accum_o, rtol=1e-3, atol=2e-5)
    seen = accum_o[..., 3] > 1e-3
    assert seen.mean() > 0.1
    alb_h = mesh.albedo.cpu().numpy()
    assert alb_h.shape == (map_size, map_size, 4) and (alb_h[..., 3] == 1).all() and alb_h.min() >= 0 and alb_h.max() <= 1
    np.testing.assert_allclose(alb_h[seen][:, :3], np.clip(alb_o[seen], 0, 1), rtol=0, atol=3e-4)
    # texels no view sees keep weight 0 before dilation; after it every texel near a chart carries some colour
    assert mesh.textureless is False


@pytest.mark.gpu
def test_edge_opposites_and_antialias_bit_exact(lib):
    """Topology table and dr.antialias restatement: same arithmetic as oracle/raster_oracle.c, no contraction -> bit-exact."""
    from mvedit_amd.mesh_ops import MeshRenderer, rasterize, antialias, edge_opposites
    from oracle import raster as OR
    from scene import icosphere
    v, f = icosphere(3, 0.6)
    v = (v * (1 + 0.3 * np.sin(6 * v[:, :1]) * np.cos(5 * v[:, 1:2]))).astype(np.float32)      # bumps: interior silhouettes
    f = f[: f.shape[0] - 40]                                                                   # a hole: boundary edges
    f = np.concatenate([f, f[:1]])                                                             # a duplicated face: non-manifold edges
    opp_o = OR.edge_opposites(f)
    t = lambda a: torch.from_numpy(a).cuda()
    assert (edge_opposites(t(f)).cpu().numpy() == opp_o).all() and (opp_o < 0).sum() > 20
    S = 96
    poses, intr = _clip_positions(v, 4, S)
    mr = MeshRenderer(near=0.01, far=100)
    v_cam, v_clip, _ = mr.project(t(v), t(poses), t(intr), S, S)
    rast = rasterize(v_clip, t(f), (S, S))
    color = np.random.default_rng(3).random((4, S, S, 8)).astype(np.float32)
    out_o = OR.antialias(color, rast.cpu().numpy(), v_clip.cpu().numpy(), f, opp_o)
    out_h = antialias(t(color), rast, v_clip, t(f)).cpu().numpy()
    changed = (out_o != color).any(-1)
    assert 0.005 < changed.mean() < 0.2, changed.mean()
    assert (out_h == out_o).all()
    # blends are convex combinations of two input pixels, applied at most four times
    assert out_o.min() >= -1e-6 and out_o.max() <= 1 + 1e-6
    out5 = antialias(t(color[..., :5].copy()), rast, v_clip, t(f)).cpu().numpy()
    assert (out5 == out_o[..., :5]).all()


@pytest.mark.gpu
@pytest.mark.parametrize('ssaa', [1, 2])
def test_mesh_renderer_forward_vs_oracle(lib, ssaa):
    """MeshRenderer.forward with a textured mesh (base_mesh_renderer.py:207-395) assembled from oracle pieces."""
    from mvedit_amd.mesh_ops import MeshRenderer, Mesh
    from oracle import raster as OR, bake_oracle as BO
    from scene import icosphere, face_atlas
    v, f = icosphere(3, 0.6)
    vn = (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)
    vt, ft = face_atlas(f)
    tex = np.random.default_rng(1).random((64, 64, 4)).astype(np.float32)
    S, nv = 64, 3
    poses, intr = _clip_positions(v, nv, S)
    t = lambda a: torch.from_numpy(a).cuda()
    mr = MeshRenderer(near=0.01, far=100, ssaa=ssaa, texture_filter='linear')
    mesh = Mesh(t(v), t(f), t(vt), t(ft), vn=t(vn), fn=t(f), albedo=t(tex))
    out = mr([mesh], t(poses)[None], t(intr)[None], S, S, dilate_edges=0, aa=True)
    assert out['rgba'].shape == (1, nv, S, S, 4) and out['depth'].shape == (1, nv, S, S) and out['normal'].shape == (1, nv, S, S, 3)
    # oracle pipeline on the projected vertices the renderer used
    Sh = S * ssaa
    v_cam, v_clip, r_c2w = mr.project(t(v), t(poses), t(intr) * ssaa, Sh, Sh)
    vc, vcl = v_cam.cpu().numpy(), v_clip.cpu().numpy()
    rast = OR.rasterize(vcl, f, (Sh, Sh))
    fg = rast[..., 3] > 0
    with np.errstate(divide='ignore'):
        depth = (1 / OR.interpolate(-vc[..., 2:3], rast, f)[..., 0]).astype(np.float32)
    depth[~fg] = 0
    nrm = OR.interpolate(vn[None], rast, f)
    nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-12)
    rot = (np.einsum('bhwk,bkj->bhwj', nrm, r_c2w.cpu().numpy()) / 2 + 0.5).astype(np.float32)
    rot[~fg] = np.array([0.5, 0.5, 1.0], np.float32)
    texc = OR.interpolate(vt[None], rast, ft)
    alb = np.stack([BO.texture_bilinear(tex[..., :3], texc[i]) for i in range(nv)])
    alb[~fg] = 0
    packed = np.concatenate([alb, fg[..., None].astype(np.float32), depth[..., None], rot], -1)
    packed = OR.antialias(packed, rast, vcl, f)
    if ssaa > 1:
        packed = packed.reshape(nv, S, ssaa, S, ssaa, 8).mean(axis=(2, 4))
    np.testing.assert_allclose(out['rgba'][0].cpu().numpy(), packed[..., :4], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out['depth'][0].cpu().numpy(), packed[..., 4], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out['normal'][0].cpu().numpy(), packed[..., 5:], rtol=0, atol=3e-6)
    a = out['rgba'][0, ..., 3]
    assert ((a > 0) & (a < 1)).float().mean() > 0.001, 'antialiasing must soften the silhouette'
    # vertex colours + shading_fun + edge dilation path
    vcol = np.concatenate([np.random.default_rng(2).random((v.shape[0], 3)), np.ones((v.shape[0], 1))], -1).astype(np.float32)
    mesh2 = Mesh(t(v), t(f), vn=t(vn), fn=t(f), vc=t(vcol))
    seen = {}

    def shade(world_pos, albedo, world_normal, fg_mask):
        seen.update(n=world_pos.shape[0], r=world_pos.norm(dim=-1).mean().item(), nn=world_normal.norm(dim=-1).mean().item())
        return albedo * 0.5
    o2 = mr([mesh2], t(poses)[None], t(intr)[None], S, S, shading_fun=shade, dilate_edges=1, aa=False)
    o3 = mr([mesh2], t(poses)[None], t(intr)[None], S, S, dilate_edges=0, aa=False)
    assert seen['n'] > 100 and abs(seen['r'] - 0.6) < 0.01 and abs(seen['nn'] - 1) < 1e-3
    m = o3['rgba'][..., 3] > (0 if ssaa == 1 else 0.999)
    np.testing.assert_allclose(o2['rgba'][..., :3][m].cpu().numpy(), 0.5 * o3['rgba'][..., :3][m].cpu().numpy(), atol=1e-6)


@pytest.mark.gpu
def test_render_ops_backward_are_exact_transposes(lib):
    """interpolate / texture / antialias are linear in their colour-like input, so the backward must be the transpose:
    <forward(x), g> == <x, backward(g)> for random x, g (fp32 accumulation order is the only difference: 1e-5 relative)."""
    from mvedit_amd.mesh_ops import MeshRenderer, rasterize, interpolate, texture, antialias
    from scene import icosphere, face_atlas
    v, f = icosphere(3, 0.6)
    v = (v * (1 + 0.25 * np.sin(6 * v[:, :1]))).astype(np.float32)
    vt, ft = face_atlas(f)
    S, nv = 96, 3
    poses, intr = _clip_positions(v, nv, S)
    t = lambda a: torch.from_numpy(a).cuda()
    mr = MeshRenderer(near=0.01, far=100)
    _, v_clip, _ = mr.project(t(v), t(poses), t(intr), S, S)
    rast = rasterize(v_clip, t(f), (S, S))
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()

    def adjoint(fwd, x):
        x = x.clone().requires_grad_(True)
        y = fwd(x)
        gy = rnd(*y.shape)
        (y * gy).sum().backward()
        lhs = (y.detach().double() * gy.double()).sum().item()
        rhs = (x.detach().double() * x.grad.double()).sum().item()
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
        assert x.grad.abs().sum() > 0
    adjoint(lambda a: interpolate(a, rast, t(f)), rnd(1, v.shape[0], 5))                     # shared attributes
    adjoint(lambda a: interpolate(a, rast, t(f)), rnd(nv, v.shape[0], 3))                    # per-view attributes
    texc = interpolate(t(vt)[None], rast, t(ft))
    adjoint(lambda tex: texture(tex, texc, rast), rnd(1, 64, 48, 3))
    adjoint(lambda col: antialias(col, rast, v_clip, t(f)), rnd(nv, S, S, 8))


@pytest.mark.gpu
@pytest.mark.parametrize('filt', ['linear', 'linear-mipmap-linear'])
def test_texture_fitting_through_mesh_renderer(lib, filt):
    """The texture pipeline's inner loop in miniature: optimise an albedo map through MeshRenderer.forward (texture fetch +
    antialias, native backward) with a stock torch optimiser until the renders match those of a ground-truth texture."""
    from mvedit_amd.mesh_ops import MeshRenderer, Mesh
    from scene import icosphere, face_atlas
    v, f = icosphere(3, 0.6)
    vn = (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)
    vt, ft = face_atlas(f)
    S, nv = 64, 6
    poses, intr = _clip_positions(v, nv, S)
    t = lambda a: torch.from_numpy(a).cuda()
    mr = MeshRenderer(near=0.01, far=100, texture_filter=filt)
    gt = torch.rand(64, 64, 3, generator=torch.Generator().manual_seed(1)).cuda()
    mk = lambda alb: Mesh(t(v), t(f), t(vt), t(ft), vn=t(vn), fn=t(f), albedo=alb)
    with torch.no_grad():
        target = mr([mk(gt)], t(poses)[None], t(intr)[None], S, S)['rgba'][..., :3]
    tex = torch.full((64, 64, 3), 0.5, device='cuda', requires_grad=True)
    opt = torch.optim.Adam([tex], lr=5e-2)
    losses = []
    for it in range(60):
        opt.zero_grad()
        out = mr([mk(tex)], t(poses)[None], t(intr)[None], S, S)['rgba'][..., :3]
        loss = ((out - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.05 * losses[0], (losses[0], losses[-1])


@pytest.mark.gpu
@pytest.mark.parametrize('filt', ['linear', 'linear-mipmap-linear'])
def test_textured_mesh_with_trainable_vertices_renders_and_backpropagates(lib, filt):
    """A textured mesh (vt + albedo) whose VERTICES require grad (base_mesh_renderer.py:240-264 with a trainable in_mesh): by default forward
    refuses (d albedo / d uv through the texture fetch is not built: an incomplete gradient must not pass silently); with
    allow_detached_uv=True it renders (the texture coordinates are treated as constants of the fetch, one warning per renderer), the texture
    and the vertices both receive finite gradients -- the vertices through rasterise / interpolate / antialias -- and the public texture() op
    still refuses a uv that requires grad."""
    import warnings
    from mvedit_amd import mesh_ops
    from mvedit_amd.mesh_ops import MeshRenderer, Mesh
    from scene import icosphere, face_atlas
    v, f = icosphere(3, 0.6)
    vn = (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)
    vt, ft = face_atlas(f)
    S, nv = 64, 4
    poses, intr = _clip_positions(v, nv, S)
    t = lambda a: torch.from_numpy(a).cuda()
    verts = t(v).clone().requires_grad_(True)
    tex = torch.rand(64, 64, 3, generator=torch.Generator().manual_seed(2)).cuda().requires_grad_(True)
    mesh = Mesh(verts, t(f), t(vt), t(ft), vn=t(vn), fn=t(f), albedo=tex)
    with pytest.raises(NotImplementedError):                                # the default: loud
        MeshRenderer(near=0.01, far=100, texture_filter=filt)([mesh], t(poses)[None], t(intr)[None], S, S)
    mr = MeshRenderer(near=0.01, far=100, texture_filter=filt, allow_detached_uv=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out = mr([mesh], t(poses)[None], t(intr)[None], S, S)['rgba']
        mr([mesh], t(poses)[None], t(intr)[None], S, S)
    assert sum('texture coordinates are detached' in str(x.message) for x in w) == 1      # once per renderer
    loss = (out[..., :3] ** 2).mean() + out[..., 3].mean()
    loss.backward()
    assert torch.isfinite(tex.grad).all() and tex.grad.abs().sum() > 0
    assert torch.isfinite(verts.grad).all() and verts.grad.abs().sum() > 0
    uv = torch.rand(1, 8, 8, 2, device='cuda', requires_grad=True)
    with pytest.raises(NotImplementedError):
        mesh_ops.texture(tex[None].detach(), uv)


@pytest.mark.gpu
def test_bake_xyz_shading_fun_and_cam_weights_uv(lib):
    """The two remaining public methods of the reference MeshRenderer (base_mesh_renderer.py:397-505) against the oracle pieces."""
    from mvedit_amd.mesh_ops import MeshRenderer, Mesh
    from oracle import raster as OR, bake_oracle as BO
    from scene import icosphere, face_atlas
    v, f = icosphere(2, 0.6)
    v = (v * (1 + 0.15 * np.sin(5 * v[:, :1]))).astype(np.float32)
    vt, ft = face_atlas(f)
    t = lambda a: torch.from_numpy(a).cuda()
    mr = MeshRenderer(near=0.01, far=100, texture_filter='linear')
    map_size = 96
    # ---- bake_xyz_shading_fun: colour = affine function of the surface position ----------------------------------------------
    mesh = Mesh(t(v), t(f), t(vt), t(ft))
    (mesh,) = mr.bake_xyz_shading_fun([mesh], lambda world_pos: world_pos * 0.5 + 0.5, map_size=map_size, dilation_iters=2)
    vt_clip = np.concatenate([vt * 2 - 1, np.tile(np.array([[0., 1.]], np.float32), (vt.shape[0], 1))], -1)[None]
    tex_rast = OR.rasterize(vt_clip, ft, (map_size, map_size))
    valid = tex_rast[0, ..., 3] > 0
    xyz = OR.interpolate(v[None], tex_rast, f)[0]
    alb = mesh.albedo.cpu().numpy()
    assert alb.shape == (map_size, map_size, 4) and (alb[..., 3] == 1).all() and mesh.textureless is False
    np.testing.assert_allclose(alb[valid][:, :3], np.clip(xyz[valid] * 0.5 + 0.5, 0, 1), rtol=0, atol=2e-6)
    assert (alb[~valid][:, :3].sum(-1) > 0).mean() > 0.3            # dilation filled texels next to the charts
    # ---- get_cam_weights_uv ----------------------------------------------------------------------------------------------------
    S, nv = 64, 5
    poses, intr = _clip_positions(v, nv, S)
    v_cam, v_clip, _ = mr.project(t(v), t(poses), t(intr), S, S)
    wts, val = mr.get_cam_weights_uv([mesh], t(poses)[None], t(intr)[None], render_size=S, map_size=map_size, render_bs=2, cos_weight_pow=2.0)
    assert wts.shape == (1, nv, map_size, map_size, 1) and (val[0].cpu().numpy() == valid).all()
    vc, vcl = v_cam.cpu().numpy(), v_clip.cpu().numpy()
    rast = OR.rasterize(vcl, f, (S, S))
    texc = OR.interpolate(vt[None], rast, ft)
    fg = rast[..., 3] > 0
    with np.errstate(divide='ignore'):
        depth = (1 / OR.interpolate(-vc[..., 2:3], rast, f)[..., 0]).astype(np.float32)
    depth[~fg] = 0
    wimg, _ = BO.view_weight(depth, np.ones((nv, S, S), np.float32), intr, 2.0)
    v_img = (vcl[..., :2] / vcl[..., 3:] * 0.5 + 0.5).astype(np.float32)
    for i in range(nv):
        vis = BO.splat_visibility(texc[i], fg[i], map_size).astype(np.float32)
        imgc = OR.interpolate(v_img[i:i + 1], tex_rast, f)[0]
        ref = BO.texture_bilinear(wimg[i][..., None], imgc)[..., 0] * vis
        np.testing.assert_allclose(wts[0, i, ..., 0].cpu().numpy(), ref, rtol=1e-3, atol=2e-5)
    assert (wts > 0).float().mean() > 0.01


# ------------------------------------------------------------------------------------------------ mip-mapped texture path
def _mip_scene(S, tex_size, nv=3, subdiv=3):
    from scene import icosphere, face_atlas
    v, f = icosphere(subdiv, 0.6)
    v = (v * (1 + 0.15 * np.sin(5 * v[:, :1]))).astype(np.float32)
    vn = (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)
    vt, ft = face_atlas(f)
    tex = np.random.default_rng(7).random((tex_size, tex_size, 4)).astype(np.float32)
    poses, intr = _clip_positions(v, nv, S)
    return v, f, vn, vt, ft, tex, poses, intr


@pytest.mark.gpu
@pytest.mark.parametrize('S,tex_size', [(64, 256), (128, 64), (256, 1024)])
def test_mip_texture_kernels_vs_oracle(lib, S, tex_size):
    """rasterize_db / interpolate_da / build_mips / texture(filter_mode='linear-mipmap-linear') and its gradient against
    oracle/texture_mip_oracle.py (nvdiffrast's algorithm restated) on the same rasterisation: minified (256^2 and 1024^2 atlases at 64^2 /
    256^2) and magnified (64^2 atlas at 128^2: level 0 only, must equal the bilinear filter)."""
    from mvedit_amd.mesh_ops import MeshRenderer, rasterize, rasterize_db, interpolate, interpolate_da, build_mips, texture
    from oracle import texture_mip_oracle as TM
    v, f, vn, vt, ft, tex, poses, intr = _mip_scene(S, tex_size)
    t = lambda a: torch.from_numpy(a).cuda()
    mr = MeshRenderer(near=0.01, far=100)
    _, v_clip, _ = mr.project(t(v), t(poses), t(intr), S, S)
    rast = rasterize(v_clip, t(f), (S, S))
    db = rasterize_db(v_clip, t(f), rast)
    db_o = TM.rasterize_db(v_clip.cpu(), torch.from_numpy(f), rast.cpu())
    # the differentials divide by a0 + a1 + a2, a sum of cross products of nearly parallel vectors for small triangles: single precision
    # leaves ~1e-4 of relative error there (nvdiffrast computes them in fp32 as well).  The bar: float64 evaluation of the same formulas is
    # the truth, the kernel may be no further from it than a few times the fp32 torch evaluation is.
    db_64 = TM.rasterize_db(v_clip.cpu().double(), torch.from_numpy(f), rast.cpu().double())
    scale = db_64.abs().max().item()
    e_ker, e_f32 = (db.cpu().double() - db_64).abs().max().item(), (db_o.double() - db_64).abs().max().item()
    assert scale > 0 and e_ker < 4 * e_f32 + 1e-6 * scale, (e_ker, e_f32, scale)
    rel = ((db.cpu().double() - db_64).abs().max(-1)[0] / (db_64.abs().max(-1)[0] + 1e-3 * scale))
    assert rel.max().item() < 5e-3, rel.max().item()
    # finite-difference sanity of the oracle itself: u at the next pixel of the SAME triangle ~ u + du/dX
    r = rast.cpu().numpy()
    same = (r[:, :, 1:, 3] == r[:, :, :-1, 3]) & (r[:, :, 1:, 3] > 0)
    du = (r[:, :, 1:, 0] - r[:, :, :-1, 0])[same]
    pred = 0.5 * (db_o[:, :, 1:, 0] + db_o[:, :, :-1, 0]).numpy()[same]
    assert np.abs(du - pred).max() < 0.05 * np.abs(du).max() + 1e-4
    texc = interpolate(t(vt)[None], rast, t(ft))
    da = interpolate_da(t(vt)[None], rast, db, t(ft))
    da_o = TM.interpolate_da(torch.from_numpy(vt), rast.cpu(), db_o, torch.from_numpy(ft))
    da_64 = TM.interpolate_da(torch.from_numpy(vt).double(), rast.cpu().double(), db_64, torch.from_numpy(ft))
    e_ker, e_f32 = (da.cpu().double() - da_64).abs().max().item(), (da_o.double() - da_64).abs().max().item()
    assert e_ker < 4 * e_f32 + 1e-6 * da_64.abs().max().item(), (e_ker, e_f32)
    # from here on the oracle runs on the KERNEL's differentials: the fetch itself is what is compared (a level fraction moves with its input)
    da_o = da.cpu()
    tx = t(tex[..., :3].copy())[None]
    mips, lv = build_mips(tx)
    lv_o = TM.build_mips(tx.cpu())
    assert lv == len(lv_o) - 1
    off = 0
    for l in range(1, lv + 1):
        n_el = lv_o[l].numel()
        assert (mips[0, off:off + n_el].cpu() - lv_o[l].reshape(-1)).abs().max().item() < 1e-6, l
        off += n_el
    out = texture(tx, texc, rast, uv_da=da, filter_mode='linear-mipmap-linear')
    ref = TM.texture(tx.cpu(), texc.cpu(), da_o) * (rast.cpu()[..., 3:] > 0)
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-5, err
    l0, l1, fr = TM.mip_level(da_o.reshape(-1, 4), tex_size, tex_size, lv)
    fgm = (rast.cpu()[..., 3] > 0).reshape(-1)
    lin = texture(tx, texc, rast, filter_mode='linear')
    if tex_size > S:          # minified: several levels in use, and the result is far from the bilinear fetch
        assert l0[fgm].max().item() >= 1 and (fr[fgm] > 0).float().mean() > 0.5
        assert (out - lin).abs().max().item() > 0.05
    else:                     # magnified: level 0 except at grazing pixels next to the silhouette; there the fetch IS the bilinear one
        mag = (fgm & (l0 == 0) & (fr == 0)).reshape(out.shape[:-1])
        assert mag.float().sum() > 0.7 * fgm.float().sum()
        assert (out.cpu() - lin.cpu())[mag].abs().max().item() < 1e-6
    # gradient w.r.t. the texture through the level stack == autograd of the oracle
    g_out = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).cuda()
    txg = tx.clone().requires_grad_(True)
    (texture(txg, texc, rast, uv_da=da, filter_mode='linear-mipmap-linear') * g_out).sum().backward()
    txo = tx.cpu().double().requires_grad_(True)
    (TM.texture(txo, texc.cpu().double(), da_o.double()) * ((rast.cpu()[..., 3:] > 0) * g_out.cpu()).double()).sum().backward()
    gs = txo.grad.abs().max().item()
    assert (txg.grad.cpu().double() - txo.grad).abs().max().item() < 2e-5 * gs + 1e-6
    # adjoint identity in isolation: <texture(T), G> == <T, texture^T(G)>
    lhs = float((out.double() * g_out.double()).sum())
    rhs = float((tx.double() * txg.grad.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * abs(lhs) + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('tag,ssaa', [('texmip_aa', 1), ('texmip_aa_ssaa2', 2)])
def test_mesh_renderer_forward_mip_mapped_vs_reference_golden(lib, tag, ssaa):
    """MeshRenderer.forward with its default texture_filter against the output of the reference's own forward EXECUTED over the mip-mapped
    stand-in dr (tests/golden/mesh_forward_ref.npz, texmip_* cases: a 256^2 atlas seen at 64^2 / 128^2)."""
    import importlib.util
    from mvedit_amd.mesh_ops import MeshRenderer, Mesh
    here = os.path.dirname(__file__)
    G = np.load(os.path.join(here, 'golden', 'mesh_forward_ref.npz'))
    spec = importlib.util.spec_from_file_location('make_mesh_forward_golden', os.path.join(here, 'golden', 'make_mesh_forward_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    v, f, vn, vt, ft, tex, vcol, poses, intr, S = mod.scene()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    mr = MeshRenderer(near=0.01, far=100, ssaa=ssaa)
    assert mr.texture_filter == 'linear-mipmap-linear'
    mesh = Mesh(t(v), t(f), t(vt), t(ft), vn=t(vn), fn=t(f), albedo=t(G['tex_big']))
    out = mr([mesh], t(poses)[None], t(intr)[None], S, S)
    # the projected vertices equal the recorded ones to rounding; the rasterisation may differ in a handful of edge pixels
    bad = (np.abs(out['rgba'][0].cpu().numpy() - G[f'{tag}_rgba']).max(-1) > 1e-4)
    assert bad.mean() < 2e-3, bad.mean()
    lin = MeshRenderer(near=0.01, far=100, ssaa=ssaa, texture_filter='linear')([mesh], t(poses)[None], t(intr)[None], S, S)
    assert (np.abs(lin['rgba'][0].cpu().numpy() - G[f'{tag}_rgba']).max(-1) > 1e-2).mean() > 0.05

