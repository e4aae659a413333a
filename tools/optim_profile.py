"""Where one nerf_optim / mesh_optim iteration spends its time (the replicated 3D update of an outer step: 96 iterations per denoise step,
lib/pipelines/mvedit_3d_pipeline.py:507-633, :716-847): bench.py's outer_step() iteration split into forward / loss / LPIPS / backward /
optimiser with a device synchronisation after each part, and the un-split iteration next to it (the difference is host overhead hidden by the
asynchronous queue).  python tools/optim_profile.py [iterations]"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from scene import sphere_density_grid  # noqa: E402
from mvedit_amd import nerf, raymarching as rm, synthetic as SY  # noqa: E402
from mvedit_amd.lpips import LPIPSEngine  # noqa: E402
from mvedit_amd.recon_loss import nerf_optim_loss  # noqa: E402
from mvedit_amd.tonemapping import Tonemapping  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(11)
meta, rows = nerf.grid_meta(12, 16, 320)
table = (torch.rand(rows, 2, generator=g) * 2 - 1) * 0.1
w1 = (torch.rand(64, 24, generator=g) * 2 - 1) * math.sqrt(6 / (64 + 24))
w2 = (torch.rand(4, 64, generator=g) * 2 - 1) * math.sqrt(6 / (4 + 64))
dec = nerf.INGPDecoderParams(table, w1, torch.zeros(64), w2, torch.tensor([2.0, 0.0, 0.0, 0.0]), 12, 320, device=dev)
bits = rm.packbits(torch.from_numpy(sphere_density_grid(128, radius=0.5)).to(dev), 0.5)
ps, P, S = 128, 1, 512
fl = S / (2 * math.tan(math.radians(15)))
intr = torch.tensor([[fl, fl, S / 2, S / 2]], device=dev)
c = torch.tensor([3.7 * math.cos(0.2), 0.0, 3.7 * math.sin(0.2)])
fwd = -c / c.norm()
right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])); right = right / right.norm()
down = torch.linalg.cross(fwd, right)
pose = torch.zeros(1, 3, 4); pose[0, :, 0], pose[0, :, 1], pose[0, :, 2], pose[0, :, 3] = right, down, fwd, c
pose = pose.to(dev)
ro, rd, _ = nerf.camera_rays(intr * (ps / S), pose, ps, ps)
ys, xs = torch.meshgrid(torch.arange(ps, dtype=torch.float32), torch.arange(ps, dtype=torch.float32), indexing='ij')
flp = ps / (2 * math.tan(math.radians(15)))
dirs = torch.stack([(xs + 0.5 - ps / 2) / flp, (ys + 0.5 - ps / 2) / flp, torch.ones_like(xs)], -1)[None].to(dev)
for t in dec.parameters().values():
    t.requires_grad_(True)
dec.max_steps = 512
vr = nerf.VolumeRenderer(dec)
vr.training = True
tm = Tonemapping(device=dev)
lights = torch.nn.functional.normalize(torch.tensor([[0.3, -0.5, -1.0]], device=dev), dim=-1)
tgt_m = torch.rand(P, ps, ps, 1, generator=g).to(dev)
tgt_rgb = torch.rand(P, ps, ps, 3, generator=g).to(dev)
lp = LPIPSEngine.from_state_dict({k: v.to(torch.bfloat16).float() for k, v in SY.make_lpips_state_dict().items()}, torch.bfloat16, device=dev)
# MVE_OPTIM_LR=0 freezes the scene (Adam does the same work, the parameters do not move): the sample count after culling stays what it is, so
# that loops timed one after the other in this process see the same workload
opt = torch.optim.Adam(list(dec.parameters().values()), lr=float(os.environ.get('MVE_OPTIM_LR', '1e-2')), eps=1e-15)
sync = torch.cuda.synchronize


def it(split, acc):
    t = [time.perf_counter()]

    def mark():
        if split:
            sync()
        t.append(time.perf_counter())
    opt.zero_grad()
    mark()
    o = vr.forward(ro, rd, bits, 128, dt_gamma=0.0)
    mark()
    res = nerf_optim_loss(o['image'], o['weights_sum'], o['depth'], o['weights'], o['ts'][0], tgt_rgb, tgt_m, dirs, torch.ones(P, device=dev),
                          lights, tonemapping=tm, shaded=True, normal_reg_weight=0.5, entropy_weight=0.2)
    mark()
    loss = res['loss'] + 0.3 * lp(res['out_rgbs'].permute(0, 3, 1, 2), tgt_rgb.permute(0, 3, 1, 2)).mean()
    mark()
    loss.backward()
    mark()
    opt.step()
    mark()
    for i in range(len(t) - 1):
        acc[i] += t[i + 1] - t[i]
    return int(o['weights'].shape[0])


if os.environ.get('MVE_OPTIM_WITH_ENGINES'):
    # the iteration inside a process that also holds the outer step's big engines (bench.py's outer_step, the real pipeline): does their presence
    # (allocator state, garbage-collector load, clocks after a heavy launch sequence) change the host-paced iteration?  Run with MVE_OPTIM_LR=0.
    import gc
    from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine
    from mvedit_amd.vae import AutoencoderKLEngine, SD_VAE_CONFIG

    def loop(n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            ns_ = it(False, [0.0] * 6)
        sync()
        return (time.perf_counter() - t0) / n * 1e3, ns_
    loop(6)
    print('before the engines exist: %.3f ms per iteration (%d samples); gc objects %d' % (*loop(N), len(gc.get_objects())), flush=True)
    unet = UNet2DConditionEngine.from_state_dict(SY.make_state_dict(dict(SD15_CONFIG), seed=1234, dtype=torch.float16), dict(SD15_CONFIG), torch.float16, 'cuda')
    xx = torch.randn(64, 4, 64, 64, device=dev, dtype=torch.float16)
    cc = torch.randn(64, 77, 768, device=dev, dtype=torch.float16)
    for _ in range(3):
        unet(xx, 499, cc)
    vae = AutoencoderKLEngine.from_state_dict(SY.make_vae_state_dict(dict(SD_VAE_CONFIG), dtype=torch.float16), dict(SD_VAE_CONFIG), torch.float16, dev)
    vae.decoder.run(xx[:8], 8)
    sync()
    print('engines resident:', round(torch.cuda.memory_allocated() / 2 ** 30, 2), 'GiB allocated,', round(torch.cuda.memory_reserved() / 2 ** 30, 2), 'GiB reserved; gc objects',
          len(gc.get_objects()), flush=True)
    loop(4)
    st0 = torch.cuda.memory_stats()
    a, ns_ = loop(N)
    st1 = torch.cuda.memory_stats()
    print(f'with engines resident: {a:.3f} ms per iteration ({ns_} samples); device mallocs during the loop {st1["num_device_alloc"] - st0["num_device_alloc"]}, '
          f'alloc retries {st1["num_alloc_retries"] - st0["num_alloc_retries"]}', flush=True)
    gc.disable()
    print('same with the garbage collector off: %.3f ms (%d samples)' % loop(N), flush=True)
    gc.enable()
    print('garbage collector on again: %.3f ms (%d samples)' % loop(N), flush=True)
    for _ in range(2):
        unet(xx, 499, cc)                      # a heavy launch sequence right before: clocks / power state
    print('right after two 64-image UNet forwards: %.3f ms (%d samples)' % loop(N), flush=True)
    del unet, vae, xx, cc
    gc.collect()
    torch.cuda.empty_cache()
    print('engines released: %.3f ms (%d samples)' % loop(N), flush=True)
    sys.exit(0)
names = ['zero_grad', 'forward (march, cull, decode, composite)', 'losses', 'LPIPS patch', 'backward', 'Adam']
for _ in range(3):
    ns = it(True, [0.0] * 6)
acc = [0.0] * 6
for _ in range(N):
    it(True, acc)
print(f'nerf_optim iteration, {ps * ps * P} rays, {ns} samples after culling; synchronised parts (ms): ' +
      ', '.join(f'{n} {a / N * 1e3:.3f}' for n, a in zip(names, acc)) + f'; sum {sum(acc) / N * 1e3:.3f}')
sync()
t0 = time.perf_counter()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(N):
    it(False, [0.0] * 6)
ev1.record()
t_host = time.perf_counter() - t0           # the host has enqueued everything; how much is still running says who is ahead
sync()
t_all = time.perf_counter() - t0
print(f'un-synchronised iteration: {t_all / N * 1e3:.3f} ms; host enqueue time {t_host / N * 1e3:.3f} ms per iteration '
      f'({"host-bound: the queue is empty when the host finishes" if t_host > 0.97 * t_all else "device-bound"}); device span {ev0.elapsed_time(ev1) / N:.3f} ms')
if os.environ.get('MVE_OPTIM_CPROFILE'):
    # where the HOST spends an iteration (the loop is host-bound: two device->host reads per iteration, as in the reference, keep the queue short)
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        it(False, [0.0] * 6)
    sync()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(45)
    st.sort_stats('tottime').print_stats(30)
if os.environ.get('MVE_OPTIM_COUNT'):
    # launches per iteration as the device sees them: kernel count from a short kineto trace
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(4):
            it(False, [0.0] * 6)
        sync()
    ev = [e for e in prof.events() if e.device_type.name == 'CUDA']
    busy = sum(e.device_time for e in ev) if ev and hasattr(ev[0], 'device_time') else sum(e.cuda_time for e in ev)
    print(f'kineto: {len(ev) / 4:.0f} device activities per iteration, device busy {busy / 4 / 1e3:.3f} ms per iteration')
    agg = {}
    for e in ev:
        a = agg.setdefault(e.name[:60], [0, 0.0]); a[0] += 1; a[1] += getattr(e, 'device_time', None) or e.cuda_time
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f'   {k:60s} x{n / 4:5.1f}  {t / 4:8.1f} us')
