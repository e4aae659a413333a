"""Per-kernel timing on the GPU (HIP events on torch's current stream).  Writes gpurun_out/microbench.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops  # noqa: E402


def timeit(fn, warm=3, it=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e-3


def main():
    dt = torch.float16
    res = []
    dev = 'cuda'
    for (M, N, K) in [(8192, 8192, 8192), (4096, 4096, 4096), (262144, 320, 320), (65536, 640, 640), (16384, 1280, 1280),
                      (262144, 2560, 320), (262144, 320, 1280), (65536, 5120, 640), (16384, 10240, 1280), (4928, 640, 768)]:
        a = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt)
        t = timeit(lambda: ops.gemm(a, w))
        res.append(dict(op='gemm', M=M, N=N, K=K, ms=t * 1e3, tflops=2 * M * N * K / t / 1e12))
        print(res[-1], flush=True)
    for (B, H, C1, Cout) in [(32, 64, 320, 320), (32, 32, 640, 640), (32, 16, 1280, 1280), (32, 8, 1280, 1280), (32, 16, 2560, 1280),
                             (32, 64, 960, 320), (8, 64, 320, 320), (8, 8, 1280, 1280)]:
        x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
        w = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt)
        t = timeit(lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64))
        res.append(dict(op='conv3x3', B=B, H=H, Cin=C1, Cout=Cout, ms=t * 1e3, tflops=2 * B * H * H * Cout * 9 * C1 / t / 1e12))
        print(res[-1], flush=True)
    for (B, L, heads, d) in [(16, 4096, 8, 40), (16, 1024, 8, 80), (16, 256, 8, 160), (8, 8192, 8, 40)]:
        C = heads * d
        qkv = torch.randn(B * L, 3 * C, device=dev, dtype=dt)
        t = timeit(lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, L, L, heads, d))
        res.append(dict(op='attention', B=B, L=L, heads=heads, d=d, ms=t * 1e3, tflops=4 * B * heads * L * L * d / t / 1e12))
        print(res[-1], flush=True)
    for (B, HW, C) in [(32, 4096, 320), (32, 1024, 640), (32, 256, 1280)]:
        x = torch.randn(B * HW, C, device=dev, dtype=dt)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        t = timeit(lambda: ops.groupnorm(x, B, HW, g, b))
        res.append(dict(op='groupnorm_silu', B=B, HW=HW, C=C, ms=t * 1e3, gbps=3 * x.numel() * 2 / t / 1e9))
        print(res[-1], flush=True)
        t = timeit(lambda: ops.layernorm(x, g, b))
        res.append(dict(op='layernorm', M=B * HW, C=C, ms=t * 1e3, gbps=2 * x.numel() * 2 / t / 1e9))
        print(res[-1], flush=True)
    # fused NeRF eval renderer: BASELINE render batch (6 views x 512^2 rays), 128^3 occupancy sphere, 12-level hash grid
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    from scene import camera_rays, sphere_density_grid
    from mvedit_amd import raymarching as G
    from mvedit_amd.nerf import INGPDecoderParams, grid_meta
    rng = np.random.default_rng(7)
    _, rows = grid_meta(12, 16, 320)
    dec = INGPDecoderParams(rng.uniform(-1e-4, 1e-4, (rows, 2)).astype(np.float32), rng.uniform(-.3, .3, (64, 24)).astype(np.float32),
                            np.zeros(64, np.float32), rng.uniform(-.3, .3, (4, 64)).astype(np.float32), np.array([2.0, 0, 0, 0], np.float32))
    bits = G.packbits(torch.from_numpy(sphere_density_grid(128, radius=0.5)).cuda(), 0.5)
    o, d = camera_rays(6, 512, seed=1, jitter=False)
    o_t, d_t = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    ws, dep, img, cnt = dec.render_rays(o_t, d_t, bits, 128, return_counts=True)
    n_samples = int(cnt.sum().item())
    t = timeit(lambda: dec.render_rays(o_t, d_t, bits, 128), warm=1, it=3)
    nbytes = o.shape[0] * 52 + n_samples * 768          # 52 B/ray + 768 B of hash-table gathers per sample
    res.append(dict(op='nerf_render_rays(fused)', rays=o.shape[0], samples=n_samples, ms=t * 1e3, views_per_s=6 / t,
                    msamples_per_s=n_samples / t / 1e6, algorithmic_gbps=nbytes / t / 1e9))
    print(res[-1], flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'microbench.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
