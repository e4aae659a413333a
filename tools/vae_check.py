"""Full-size check + timing of the native SD VAE (run on the GPU box): 512 x 512 decode / encode of a batch of views, per-class
time split, chunk invariance, finiteness.  python tools/vae_check.py [views] [--detail]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from mvedit_amd import synthetic as SY
from mvedit_amd.vae import AutoencoderKLEngine, SD_VAE_CONFIG

V = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
dev = torch.device('cuda')
cfg = dict(SD_VAE_CONFIG)
t0 = time.time()
eng = AutoencoderKLEngine.from_state_dict(SY.make_vae_state_dict(cfg, dtype=torch.float16), cfg, torch.float16, dev)
print(f'weights {time.time() - t0:.1f}s', flush=True)
g = torch.Generator().manual_seed(0)
z = torch.randn(V, 4, 64, 64, generator=g).to(dev, torch.float16)
x = (torch.rand(V, 3, 512, 512, generator=g) * 2 - 1).to(dev, torch.float16)


def timed(fn, it=2):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


for name, half, inp in (('decode', eng.decoder, z), ('encode', eng.encoder, x)):
    out = half.run(inp, 8)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all(), name
    parts = half.run(inp, 4)
    print(name, 'out', tuple(out.shape), 'std %.3f' % out.float().std().item(), 'chunk(8) == chunk(4):', torch.equal(out, parts), flush=True)
    ms = timed(lambda: half.run(inp, 8))
    fl = sum(half.plan(min(V, 8), inp.shape[2], inp.shape[3], torch.float16)['flops'].values()) / min(V, 8) * V
    print(f'{name}: {V} views {ms:.1f} ms = {ms / V:.2f} ms/view, {fl / ms / 1e9:.0f} TFLOP/s', flush=True)
    _, prof = half.run(inp[:min(V, 8)], 8, profile=True)
    agg = {}
    for c, lab, f, m in prof[0]:
        a = agg.setdefault((c, lab), [0.0, 0.0, 0])
        a[0] += m; a[1] += f; a[2] += 1
    for (c, lab), (m, f, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f'   {c:9s} {lab:34s} x{n:3d} {m:8.2f} ms  {f / max(m, 1e-9) / 1e9:7.0f} TF/s')
    if '--detail' in sys.argv:          # every op in execution order: which conv shapes run below the class average
        for c, lab, f, m in prof[0]:
            if m > 0.05:
                print(f'      {c:9s} {lab:34s} {m:7.3f} ms  {f / 1e9:9.1f} GFLOP  {f / max(m, 1e-9) / 1e9:7.0f} TF/s')
