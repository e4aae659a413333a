"""Writes tests/golden/volume_renderer_ref.npz by EXECUTING the reference's `VolumeRenderer.forward`, inference branch
(lib/models/decoders/base_volume_renderer.py:179-343; the method and `batch_near_far_from_aabb` of lib/ops/raymarching/raymarching.py
are taken from the files with `ast`).  The CUDA operators it calls (near_far_from_aabb, march_rays, composite_rays) and the decoder
are replaced by the CPU oracle's functions -- which are themselves pinned against the reference's own kernels
(tests/test_raymarching_ref.py) -- so what this file pins is the ORCHESTRATION the repo restates in oracle/nerf_oracle.py:render_rays_eval:
the n_step schedule, the alive-list compaction, the termination rule, the in-place accumulation.
Run from the repo root (needs /root/reference):  python tests/golden/make_volume_renderer_golden.py"""
import ast
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import nerf_oracle as N  # noqa: E402
from oracle import raymarching as ORM  # noqa: E402
from scene import sphere_density_grid  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'volume_renderer_ref.npz')


def _fn(path, name, ns, cls=None):
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    node = [n for n in body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    exec(compile(ast.Module([node], []), path, 'exec'), ns)
    return ns[name]


def density_fn(x):
    """analytic stand-in for the decoder's density (same on both sides of the comparison)"""
    return 40.0 * torch.exp(-8.0 * (x * x).sum(-1)) * (1.0 + 0.3 * torch.sin(9.0 * x[..., 0]))


def scene(S=20, G=32, seed=3):
    params = N.make_nerf_params(seed=seed, table_scale=0.5)           # a table large enough for visible structure
    grid = sphere_density_grid(G, radius=0.6)
    bits = ORM.packbits(grid, 0.5)
    f = S / (2 * np.tan(np.radians(20)))
    intr = np.array([[f, f, S / 2, S / 2]], np.float32)
    pose = np.eye(4, dtype=np.float32)[None, :3]
    pose[0, :, 3] = (0.1, -0.05, -2.2)
    dirs = N.get_ray_directions(S, S, intr)
    rays_o, rays_d = N.get_rays(dirs, pose)
    return params, bits.reshape(-1), G, rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)


def main():
    params, bits, G, rays_o, rays_d = scene()
    t = torch.from_numpy

    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        n, f = ORM.near_far_from_aabb(rays_o.numpy(), rays_d.numpy(), aabb.numpy(), min_near)
        return t(n), t(f)

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, perturb=False, dt_gamma=0,
                   max_steps=1024, contract=False):
        assert not perturb
        x, d, ts = ORM.march_rays(n_alive, n_step, rays_alive.numpy(), rays_t.numpy(), rays_o.numpy(), rays_d.numpy(), bound,
                                  density_bitfield.numpy(), C, H, near.numpy(), far.numpy(), np.zeros(n_alive, np.float32), dt_gamma, max_steps)
        return t(x), t(d), t(ts)

    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh=1e-2, binarize=False):
        # the numpy views share the tensors' storage: in place, like the reference's operator
        ORM.composite_rays(n_alive, n_step, rays_alive.numpy(), rays_t.numpy(), sigmas.numpy(), rgbs.numpy(), ts.numpy(), weights_sum.numpy(),
                           depth.numpy(), image.numpy(), T_thresh, binarize)

    def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, perturb=False, dt_gamma=0, max_steps=1024, contract=False):
        assert not perturb
        x, d, ts, rays = ORM.march_rays_train(rays_o.numpy(), rays_d.numpy(), bound, density_bitfield.numpy(), C, H, nears.numpy(), fars.numpy(),
                                              np.zeros(rays_o.shape[0], np.float32), dt_gamma, max_steps)
        return t(x), t(d), t(ts), t(rays)

    def composite_rays_train(sigmas, rgbs, ts, rays, T_thresh=1e-4, binarize=False):
        return tuple(t(a) for a in ORM.composite_rays_train(sigmas.numpy(), rgbs.numpy(), ts.numpy(), rays.numpy(), T_thresh, binarize))

    ns = dict(torch=torch, F=F, near_far_from_aabb=near_far_from_aabb, march_rays=march_rays, composite_rays=composite_rays,
              march_rays_train=march_rays_train, composite_rays_train=composite_rays_train)
    _fn(os.path.join(REF, 'lib/ops/raymarching/raymarching.py'), 'batch_composite_rays_train', ns)
    _fn(os.path.join(REF, 'lib/ops/raymarching/raymarching.py'), 'batch_near_far_from_aabb', ns)
    forward = _fn(os.path.join(REF, 'lib/models/decoders/base_volume_renderer.py'), 'forward', ns, cls='VolumeRenderer')

    class Renderer:
        training = False
        bound, min_near, max_steps, pre_gamma, post_gamma = 1.0, 0.2, 256, None, None
        aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])

        def preproc(self, code):
            return code

        def point_decode(self, xyzs, dirs, code):
            s, c = N.point_decode(xyzs[0].detach().numpy(), params, 1.0)
            return t(s), t(c), [len(s)]

    out = {}
    for tag, dtg in (('plain', 0.0), ('dt_gamma', 1.0 / 64)):
        res = forward(Renderer(), t(rays_o)[None], t(rays_d)[None], [None], t(bits)[None], G, dt_gamma=torch.tensor([dtg]))
        out[f'{tag}_weights_sum'], out[f'{tag}_depth'], out[f'{tag}_image'] = (res[k][0].numpy() for k in ('weights_sum', 'depth', 'image'))
    # training branch (:207-262): two-pass march, weight culling, decode, differentiable compositing (forward values only)
    class TrainRenderer(Renderer):
        training = True
        weight_culling_th = 1e-3

        def point_density_decode(self, xyzs, code):
            s, _ = N.point_decode(xyzs[0].detach().numpy(), params, 1.0)
            return t(s), [len(s)]

        def point_decode(self, xyzs, dirs, code, use_2nd_order=False):
            s, c = N.point_decode(xyzs[0].detach().numpy(), params, 1.0)
            return t(s), t(c), [len(s)]
    res = forward(TrainRenderer(), t(rays_o)[None], t(rays_d)[None], [None], t(bits)[None], G, dt_gamma=0.0)
    out['train_weights'] = res['weights'].numpy()
    out['train_weights_sum'], out['train_depth'], out['train_image'] = (res[k][0].numpy() for k in ('weights_sum', 'depth', 'image'))
    out['train_rays'], out['train_ts'] = res['rays'][0].numpy(), res['ts'][0].numpy()
    # BaseNeRF.render (lib/models/autoencoders/base_nerf.py:489-556) with cfg = dict(return_rgba, compute_normal), calling the forward
    # above as its decoder and the reference's own geometry helpers (lib/core/utils/geometry_utils.py)
    gns = dict(torch=torch, F=F, np=np)
    for name in ('get_ray_directions', 'get_rays', 'depth_to_normal'):
        _fn(os.path.join(REF, 'lib/core/utils/geometry_utils.py'), name, gns)
    rns = dict(torch=torch, get_ray_directions=gns['get_ray_directions'], get_rays=gns['get_rays'], depth_to_normal=gns['depth_to_normal'])
    render = _fn(os.path.join(REF, 'lib/models/autoencoders/base_nerf.py'), 'render', rns, cls='BaseNeRF')

    class Decoder(Renderer):
        def train(self, mode=True):
            self.training = mode

        def __call__(self, *a, **k):
            return forward(self, *a, **k)

    class NeRF:
        bg_color, grid_size = 1.0, G
    S, b = 16, 2
    f = S / (2 * np.tan(np.radians(20)))
    intr = torch.tensor([[f, 1.05 * f, S / 2 + 0.3, S / 2 - 0.2], [0.9 * f, f, S / 2, S / 2]], dtype=torch.float32)
    poses = torch.eye(4)[None, :3].repeat(b, 1, 1)
    poses[0, :, 3] = torch.tensor([0.1, -0.05, -2.2])
    c, sn = np.cos(0.6), np.sin(0.6)
    poses[1, :, :3] = torch.tensor([[c, 0, sn], [0, 1, 0], [-sn, 0, c]], dtype=torch.float32)
    poses[1, :, 3] = torch.tensor([-2.2 * sn, 0.0, -2.2 * c])
    rgba, depth, normal, normal_fg = render(NeRF(), Decoder(), [None], t(bits)[None], S, S, intr[None], poses[None],
                                            cfg=dict(return_rgba=True, compute_normal=True, dt_gamma_scale=0.5))
    out.update(render_intrinsics=intr.numpy(), render_poses=poses.numpy(), render_rgba=rgba[0].numpy(), render_depth=depth[0].numpy(),
               render_normal=normal[0].numpy(), render_normal_fg=normal_fg[0].numpy())
    # update_extra_state (:105-177): full refresh (iter_density < 16) of the density grid + bitfield, one scene.
    # Random draws come from torch's CPU generator after manual_seed, in the method's own order; tests/test_nerf_ref.py replays them.
    def packbits(grid, thresh, bitfield=None):
        res = t(ORM.packbits(grid.numpy(), float(thresh)))
        if bitfield is not None:
            bitfield.view(-1)[:] = res
            return bitfield
        return res
    uns = dict(torch=torch, get_module_device=lambda m: torch.device('cpu'), custom_meshgrid=lambda *a: torch.meshgrid(*a, indexing='ij'),
               morton3D=lambda c: t(ORM.morton3D(c.numpy())), morton3D_invert=lambda i: t(ORM.morton3D_invert(i.int().numpy())), packbits=packbits)
    update = _fn(os.path.join(REF, 'lib/models/decoders/base_volume_renderer.py'), 'update_extra_state', uns, cls='VolumeRenderer')

    class GridRenderer(Renderer):
        def point_density_decode(self, xyzs, code):
            x = xyzs.reshape(-1, 3)
            return density_fn(x), [x.shape[0]]
    H = 16
    # (only the full refresh: the pipelines always pass iter_density = 0 -- mvedit_3d_pipeline.py:496-510 never advances it -- and the
    # partial branch of the method raises a shape error for one scene: its [2, N] index meets a [1, 2N] value at :163)
    for tag, it, seed in (('full', 0, 11),):
        grid0 = torch.zeros(1, H ** 3)
        bitfield = torch.zeros(1, H ** 3 // 8, dtype=torch.uint8)
        torch.manual_seed(seed)
        update(GridRenderer(), [None], grid0, bitfield, it, density_thresh=0.01, decay=0.9, S=128)
        out[f'grid_{tag}_after'], out[f'grid_{tag}_bits'] = grid0.numpy(), bitfield.numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: (v.shape, float(np.abs(v).mean())) for k, v in out.items()})


if __name__ == '__main__':
    main()
