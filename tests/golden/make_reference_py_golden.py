"""Generates tests/golden/reference_py.npz by EXECUTING the reference's own pure-torch helpers in this
container (they cannot travel to the GPU box, the vectors can):

  lib/core/utils/geometry_utils.py : get_ray_directions, get_rays, depth_to_normal, normalize_depth
      (the module imports mcubes / skimage, which are not installed, so the four function definitions are
       extracted from the file with `ast` and executed verbatim in a namespace holding torch / F / np)
  lib/core/utils/camera_utils.py   : look_at, random_surround_views (same extraction)
  lib/ops/edge_dilation.py         : edge_dilation (imported as a module: it only needs torch)
  lib/core/diffusion.py            : get_noise_scales (extracted)
  lib/pipelines/utils.py           : get_camera_dists, prune_cameras (extracted, with lib/ops/rotation_conversions.py helpers)

Nothing is copied into the repo; the reference sources are read where they lie under /root/reference.
Run from the repo root (needs /root/reference):  python tests/golden/make_reference_py_golden.py
"""
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree
import ast
import importlib.util
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_py.npz')


def extract(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = dict(torch=torch, F=F, np=np, math=math)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, 'exec'), ns)
    return [ns[n] for n in names]


def main():
    get_ray_directions, get_rays, depth_to_normal, normalize_depth = extract(
        os.path.join(REF, 'lib/core/utils/geometry_utils.py'),
        ['get_ray_directions', 'get_rays', 'depth_to_normal', 'normalize_depth'])
    look_at, random_surround_views = extract(os.path.join(REF, 'lib/core/utils/camera_utils.py'),
                                             ['look_at', 'random_surround_views'])
    spec = importlib.util.spec_from_file_location('ref_edge_dilation', os.path.join(REF, 'lib/ops/edge_dilation.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    out = {}
    torch.manual_seed(0)
    poses = random_surround_views(3.7, 8, -0.3, 0.6, use_linspace=True)           # lib/apis/adapter3d.py:991-994
    out['poses'] = poses.numpy()
    h = w = 24
    f = h / (2 * math.tan(math.radians(15)))
    intr = torch.tensor([[f, f * 1.1, w / 2 + 0.7, h / 2 - 0.3]] * 8, dtype=torch.float32)
    intr[:, 0] *= torch.linspace(0.9, 1.1, 8)
    out['intrinsics'] = intr.numpy()
    dirs = get_ray_directions(h, w, intr[None], norm=False)
    rays_o, rays_d = get_rays(dirs, poses[None, :, :3], norm=True)
    out['directions'] = dirs[0].numpy()
    out['rays_o'] = rays_o[0].numpy()
    out['rays_d'] = rays_d[0].numpy()
    g = torch.Generator().manual_seed(1)
    depth = 0.2 + 0.3 * torch.rand(8, h, w, generator=g)
    depth[0, :4] = 0.0                                                             # background: clamp(min=1e-6) path
    out['depth_in'] = depth.numpy()
    out['normal'] = depth_to_normal(depth, dirs[0]).numpy()
    alphas = torch.rand(8, h, w, 1, generator=g)
    alphas[1, 5:9] = 0.0
    out['alphas_in'] = alphas.numpy()
    out['depth_norm'] = normalize_depth(depth * alphas.squeeze(-1), alphas).numpy()
    # edge dilation: a texture atlas with a ragged valid region
    img = torch.rand(2, 3, 40, 48, generator=g)
    yy, xx = torch.meshgrid(torch.arange(40), torch.arange(48), indexing='ij')
    mask = (((yy - 20) ** 2 + (xx - 22) ** 2) < 150) | ((yy > 30) & (xx % 7 < 3))
    mask = mask[None, None].float().expand(2, 1, -1, -1).clone()
    mask[1] = (torch.rand(1, 40, 48, generator=g) > 0.8).float()
    img = img * mask
    out['dil_img'] = img.numpy()
    out['dil_mask'] = mask.numpy()
    out['dil_out_r3_i7'] = mod.edge_dilation(img, mask, radius=3, iters=7).numpy()
    out['dil_out_r1_i2'] = mod.edge_dilation(img, mask, radius=1, iters=2).numpy()
    # camera pruning bookkeeping (lib/pipelines/utils.py:350-379; needs matrix_to_quaternion from lib/ops/rotation_conversions.py)
    ns = dict(torch=torch, F=F, np=np, math=math)
    for path, names in ((os.path.join(REF, 'lib/ops/rotation_conversions.py'), ['_sqrt_positive_part', 'matrix_to_quaternion']),
                        (os.path.join(REF, 'lib/pipelines/utils.py'), ['get_camera_dists', 'prune_cameras'])):
        tree = ast.parse(open(path).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                exec(compile(ast.Module([node], []), path, 'exec'), ns)
    torch.manual_seed(3)
    poses32 = random_surround_views(3.7, 32, -0.3, 0.6, use_linspace=True)
    poses32 = poses32[torch.randperm(32)]
    cw = 0.5 + torch.rand(32)
    pd = torch.rand(32)
    out['prune_poses'] = poses32.numpy()
    out['prune_cam_weights'] = cw.numpy()
    out['prune_pixel_dist'] = pd.numpy()
    d = ns['get_camera_dists'](poses32, cw, 'cpu')
    out['prune_dists'] = d.numpy()
    k1, d1 = ns['prune_cameras'](d.clone(), 1, 16, 'cpu')
    k2, d2 = ns['prune_cameras'](d.clone(), 4, 9, 'cpu', pixel_dist=pd.clone())
    out['prune_keep_16'], out['prune_dists_16'] = k1.numpy(), d1.numpy()
    out['prune_keep_9'], out['prune_dists_9'] = k2.numpy(), d2.numpy()
    # noise scales of integer / fractional timesteps (lib/core/diffusion.py:4-21)
    (gns,) = extract(os.path.join(REF, 'lib/core/diffusion.py'), ['get_noise_scales'])
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2      # SD's scaled-linear schedule
    ab = torch.cumprod(1 - betas, dim=0)
    out['ns_alphas_bar'] = ab.numpy()
    ti = torch.tensor([0, 1, 17, 500, 998, 999])
    tf = torch.tensor([0.0, 0.25, 17.5, 499.999, 998.75, 999.0])
    out['ns_t_int'], out['ns_t_float'] = ti.numpy(), tf.numpy()
    a, b = gns(ab, ti, 1000)
    out['ns_int_a'], out['ns_int_b'] = a.numpy(), b.numpy()
    a, b = gns(ab, tf, 1000)
    out['ns_float_a'], out['ns_float_b'] = a.numpy(), b.numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes;', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
