"""Experiment: lane utilisation and ray ordering (row-major vs 8x8 tiles) for the fused NeRF renderer."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from scene import sphere_density_grid
from mvedit_amd import nerf, raymarching as rm
from tools.microbench import timeit
dev = torch.device('cuda', 0)
S, nv = 512, 6
f = S / (2 * math.tan(math.radians(15)))
intr = torch.tensor([[f, f, S / 2, S / 2]] * nv, device=dev)
poses = bench.surround_poses(nv).to(dev)
meta, rows = nerf.grid_meta(12, 16, 320)
g = torch.Generator().manual_seed(7)
dec = nerf.INGPDecoderParams((torch.rand(rows, 2, generator=g) * 2 - 1) * 1e-4, (torch.rand(64, 24, generator=g) - 0.5) * 0.5, torch.zeros(64),
                             (torch.rand(4, 64, generator=g) - 0.5) * 0.5, torch.tensor([2.0, 0, 0, 0]), 12, 320, device=dev)
bits = rm.packbits(torch.from_numpy(sphere_density_grid(128, radius=0.5)).to(dev), 0.5)
ro, rd, _ = nerf.camera_rays(intr, poses, S, S)
_, _, _, cnt = dec.render_rays(ro, rd, bits, 128, 0.0, return_counts=True)
def util(c):
    w = c.view(-1, 64).float()
    return (w.sum() / (64 * w.max(dim=1).values.sum())).item()
print('row-major: lane utilisation', round(util(cnt), 3), 'samples', int(cnt.sum()))
t0 = timeit(lambda: dec.render_rays(ro, rd, bits, 128, 0.0), 2, 5) * 1e3
# 8x8 tiles: ray index -> (view, ty, tx, y8, x8)
idx = torch.arange(nv * S * S, device=dev).view(nv, S // 8, 8, S // 8, 8).permute(0, 1, 3, 2, 4).reshape(-1)
ro2, rd2 = ro[idx].contiguous(), rd[idx].contiguous()
_, _, _, cnt2 = dec.render_rays(ro2, rd2, bits, 128, 0.0, return_counts=True)
print('8x8 tiles: lane utilisation', round(util(cnt2), 3))
t1 = timeit(lambda: dec.render_rays(ro2, rd2, bits, 128, 0.0), 2, 5) * 1e3
print(f'render_rays only: row-major {t0:.2f} ms, 8x8 tiles {t1:.2f} ms')
# only the rays that hit: upper bound of a perfect scheduler
hit = cnt > 0
roh, rdh = ro[hit].contiguous(), rd[hit].contiguous()
t2 = timeit(lambda: dec.render_rays(roh, rdh, bits, 128, 0.0), 2, 5) * 1e3
print(f'hit rays only ({int(hit.sum())} of {hit.numel()}): {t2:.2f} ms')
