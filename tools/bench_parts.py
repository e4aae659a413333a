"""The parts of the benchmark line that are not the headline measurement: power / clock probes, the secondary figures (render, raster, bake, VAE,
ControlNet, TRACER ...), the compact extra workloads, the outer step of the reference loop composed from the engine's objects, the strong-scaling
projection.  bench.py (the driver's contract: headline step, roofline, cpu_baseline, parity) imports them; nothing here touches oracle/ --
the checker lives in bench.py's cpu_baseline alone (tests/test_abi.py)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS_F16 = 2500.0      # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
VIEWS = 32
LATENT = 64
CTX_LEN = 77
GUIDANCE = 7.0


class PowerSampler:
    """Socket power and shader clock from rocm-smi while the timed region runs (the MFMA-heavy kernels sit at the chip's power cap: the
    clock they sustain, not the instruction schedule, sets their wall time -- DESIGN.md section 5)."""

    def __init__(self, period=0.1):
        import threading
        self.period, self.samples, self._stop = period, [], False
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
                p = re.search(r'Power \(W\):\s*([\d.]+)', out)
                c = re.search(r'sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)', out)
                if p and c:
                    self.samples.append((float(p.group(1)), int(c.group(1))))
            except Exception:
                return
            time.sleep(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self._th.join(timeout=6)

    def summary(self):
        if not self.samples:
            return None
        n = len(self.samples)
        return dict(avg_w=round(sum(s[0] for s in self.samples) / n, 1), max_w=round(max(s[0] for s in self.samples), 1),
                    avg_sclk_mhz=round(sum(s[1] for s in self.samples) / n), samples=n, source='rocm-smi --showpower --showclocks over ~2.5 s of extra untimed steps right after the timed region')


def pmc_traffic(cls):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when no profile of this kernel is committed."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        rec = json.load(open(path)).get(cls)
        if not rec or rec.get('fetch_kib_mean') is None or rec.get('write_kib_mean') is None:
            return None
        return float((2 * rec['fetch_kib_mean'] + rec['write_kib_mean']) * 1024)
    except (OSError, ValueError):
        return None


def clock_probe(dev):
    """Shader clock the chip sustains inside the dominant conv kernel (DVFS: the MFMA-heavy kernels run far below the 2.4 GHz the
    2.5 PFLOP/s peak is quoted at).  One instrumented launch of the ping-pong conv kernel (mve_gemm_pp_profile: every wave records
    s_memtime at kernel entry / exit) on a UNet level-1 shape; clock = cycles of the longest wave / HIP-event wall time."""
    import ctypes
    from mvedit_amd import ops, _lib
    prof = _lib.raw('mve_gemm_pp_profile')
    prof.argtypes = [ctypes.c_void_p]
    B, H, C = 64, 32, 640
    x = torch.randn(B * H * H, C, device=dev, dtype=torch.float16)
    w = torch.randn(C, C // 64, 3, 3, 64, device=dev, dtype=torch.float16) * (9 * C) ** -0.5
    f = lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=False)
    nblocks = (B * H * H // 256) * (C // 320)
    buf = torch.zeros(nblocks * 64, dtype=torch.int64, device=dev)
    try:
        prof(ctypes.c_void_p(buf.data_ptr()))
        f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            f()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
    finally:
        prof(None)
    v = buf.view(nblocks, 8, 8).double().cpu()
    per_tile = float((v[:, :, :4].sum(-1) + v[:, :, 6] + v[:, :, 7]).mean())          # prologue + K loop + epilogue cycles of a tile
    tiles_per_cu = (nblocks + 255) // 256
    ghz = per_tile * tiles_per_cu / (ms * 1e-3) / 1e9
    return dict(shader_ghz=round(ghz, 3), quoted_ghz=2.4, peak_at_clock=round(PEAK_TFLOPS_F16 * ghz / 2.4, 1),
                probe='k_gemm_pp<MODE=1>, 64 x 32 x 32 x 640 -> 640 conv, s_memtime per wave / HIP-event wall time (lower bound: launch gaps count as cycles)')


def surround_poses(n, radius=3.7, elev=0.2):
    """n look-at c2w matrices (OpenCV convention: x right, y down, z forward) on a circle, as the reference's camera rig
    (lib/apis/adapter3d.py:991-996: distance 3.7, fov 30 degrees)."""
    import math
    poses = torch.zeros(n, 3, 4)
    for i in range(n):
        az = 2 * math.pi * i / n
        c = torch.tensor([radius * math.cos(elev) * math.cos(az), radius * math.cos(elev) * math.sin(az), radius * math.sin(elev)])
        fwd = -c / c.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        poses[i, :, 0], poses[i, :, 1], poses[i, :, 2], poses[i, :, 3] = right, down, fwd, c
    return poses


def secondary(dev):
    """SURVEY section 8(d) secondary figures on rank 0: NeRF render views/s, raster views/s, back-projection texel-views/s, each
    with its algorithmic HBM rate.  Synthetic scene: 128^3 occupancy sphere + 12-level hash grid; 81,920-face icosphere."""
    import math
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from scene import face_atlas, icosphere, sphere_density_grid
    from mvedit_amd import nerf, raymarching as rm
    from mvedit_amd.mesh_ops import Mesh, MeshRenderer, rasterize

    def timed(fn, it=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e-3

    out = {}
    S, nv = 512, 6
    f = S / (2 * math.tan(math.radians(15)))
    intr = torch.tensor([[f, f, S / 2, S / 2]] * nv, device=dev)
    poses = surround_poses(nv).to(dev)
    # ---- NeRF: fused march + hash grid + MLP + composite (BaseNeRF.render, one launch) ----------------------------------
    meta, rows = nerf.grid_meta(12, 16, 320)
    g = torch.Generator().manual_seed(7)
    table = (torch.rand(rows, 2, generator=g) * 2 - 1) * 1e-4                       # ingp_decoder.py:88 init
    w1 = (torch.rand(64, 24, generator=g) * 2 - 1) * math.sqrt(6 / (64 + 24))
    w2 = (torch.rand(4, 64, generator=g) * 2 - 1) * math.sqrt(6 / (4 + 64))
    dec = nerf.INGPDecoderParams(table, w1, torch.zeros(64), w2, torch.tensor([2.0, 0.0, 0.0, 0.0]), 12, 320, device=dev)
    bits = rm.packbits(torch.from_numpy(sphere_density_grid(128, radius=0.5)).to(dev), 0.5)
    ro, rd, _ = nerf.camera_rays(intr, poses, S, S)
    _, _, _, cnt = dec.render_rays(ro, rd, bits, 128, 0.0, return_counts=True)
    samples = int(cnt.sum().item())
    nr = nerf.NeRFRenderer(grid_size=128)
    cfg = dict(return_rgba=True, compute_normal=True, dt_gamma_scale=0.0)
    t = timed(lambda: nr.render(dec, None, bits[None], S, S, intr[None], poses[None], cfg=cfg))
    nbytes = ro.shape[0] * 52 + samples * (56 + 768)
    # The renderer is bound by the rate at which a CU's texture path resolves lane addresses, not by HBM bytes: a sample is 12 levels x 8 corner
    # fetches of 8 bytes from a ~30 MB table.  tools/probe/gather_probe.hip (profiles/r03_gather_probe.log) measures what the chip sustains on
    # UNRELATED 8-byte fetches -- 268 G/s from a 2 MiB (L2-resident) table, 80 G/s from 16 MiB (LLC), 56 G/s from HBM, independent of occupancy
    # and of the fetches in flight per lane (a throughput limit: ~0.5 lane addresses per CU per clock).  The renderer's fetches are not unrelated
    # (corner pairs share a line, neighbouring rays share cells), which is the only reason it can exceed those figures.
    fetches = samples * 96
    out['nerf_render'] = dict(views_per_s=round(nv / t, 1), ms=round(t * 1e3, 2), rays=ro.shape[0], samples=samples,
                              algorithmic_GBps=round(nbytes / t / 1e9, 1),
                              fetch_model=dict(fetches_per_sample=96, fetch_rate_G_per_s=round(fetches / t / 1e9, 1),
                                               random_fetch_rate_G_per_s=dict(l2_2MiB=268.0, llc_16MiB=80.0, hbm_1GiB=56.0),
                                               vs_random_llc=round(fetches / t / 80e9, 2), source='profiles/r03_gather_probe.log'))
    # ---- mesh: rasterise + full MeshRenderer.forward -------------------------------------------------------------------------
    v, fc = icosphere(6, 0.6)
    vt, ft = face_atlas(fc)
    tv = lambda a: torch.from_numpy(a).to(dev)
    vn = tv((v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32))
    mr = MeshRenderer(near=0.01, far=100)
    _, v_clip, _ = mr.project(tv(v), poses, intr, S, S)
    faces = tv(fc)
    t = timed(lambda: rasterize(v_clip, faces, (S, S)))
    rb = nv * S * S * 16 + v.shape[0] * 36 * nv + fc.shape[0] * 12 * nv
    out['rasterize'] = dict(views_per_s=round(nv / t, 1), ms=round(t * 1e3, 3), faces=int(fc.shape[0]), algorithmic_GBps=round(rb / t / 1e9, 1))
    mesh = Mesh(tv(v), faces, tv(vt), tv(ft), vn=vn, fn=faces, albedo=torch.rand(1024, 1024, 4, device=dev))
    t = timed(lambda: mr([mesh], poses[None], intr[None], S, S))
    out['mesh_forward'] = dict(views_per_s=round(nv / t, 1), ms=round(t * 1e3, 3))
    # ---- back-projection: 32 views 512^2 -> 1024^2 atlas ------------------------------------------------------------------------
    V = 32
    poses32 = surround_poses(V).to(dev)
    intr32 = intr[:1].expand(V, -1).contiguous()
    images = torch.rand(1, V, S, S, 3, device=dev)
    alphas = torch.ones(1, V, S, S, 1, device=dev)
    t = timed(lambda: mr.bake_multiview([mesh], images, alphas, poses32[None], intr32[None], map_size=1024, render_bs=8), it=2)
    tb = V * 1024 * 1024 * 36
    out['bake_multiview'] = dict(texel_views_per_s=round(V * 1024 * 1024 / t / 1e9, 3), unit='G texel-views/s', ms=round(t * 1e3, 2),
                                 algorithmic_GBps=round(tb / t / 1e9, 1),
                                 note='atomic- and gather-rate bound (64-bit visibility atomics, mip-mapped fetches): not priced against HBM bytes')
    # ---- one SD-1.5 ControlNet over the 64 images of a step (0.28 TFLOP per image incl. the 512^2 embedding) ----
    from mvedit_amd.controlnet import ControlNetEngine
    from mvedit_amd.unet import SD15_CONFIG
    from mvedit_amd import synthetic as SY
    cn = ControlNetEngine.from_state_dict(SY.make_controlnet_state_dict(dict(SD15_CONFIG), dtype=torch.float16), dict(SD15_CONFIG), torch.float16, dev)
    Bc = 2 * VIEWS
    xs = torch.randn(Bc, 4, LATENT, LATENT, device=dev, dtype=torch.float16)
    cs = torch.randn(Bc, CTX_LEN, 768, device=dev, dtype=torch.float16)
    ci = torch.rand(Bc, 3, 8 * LATENT, 8 * LATENT, device=dev, dtype=torch.float16)
    dn, md = cn.new_outputs(Bc, LATENT, LATENT)
    t = timed(lambda: cn.run(xs, 499, cs, ci, 1.0, dn, md, False), it=2)
    fl = sum(cn.plan(Bc, LATENT, LATENT, CTX_LEN)['flops'][k] for k in ('conv3x3', 'linear', 'attention'))
    out['controlnet_forward'] = dict(ms=round(t * 1e3, 2), images=Bc, tflops_per_s=round(fl / t / 1e12, 1))
    del cn, dn, md
    # ---- DMTet on the 128^3 tet grid of the reference's mesh stage (6 tets per cube, 12.6 M tets) ----------------------------------
    from scene import tet_grid
    from mvedit_amd.mesh_ops import DMTet
    pos, tets = tet_grid(128)
    tp, tt = tv(pos), torch.from_numpy(tets).to(dev)
    sdf = (0.6 - tp.norm(dim=-1) + 0.02 * torch.sin(9 * tp[:, 0]) * torch.cos(7 * tp[:, 1])).contiguous()
    dm = DMTet(dev)
    vv, ff = dm(tp, sdf, tt)
    t = timed(lambda: dm(tp, sdf, tt))
    db = tets.shape[0] * (16 + 8 + 8) + pos.shape[0] * 24
    out['dmtet'] = dict(ms=round(t * 1e3, 3), tets=int(tets.shape[0]), verts_out=int(vv.shape[0]), faces_out=int(ff.shape[0]),
                        mtets_per_s=round(tets.shape[0] / t / 1e6, 1), algorithmic_GBps=round(db / t / 1e9, 1))
    # ---- SD VAE: decode the V views' x0 latents to 512^2 images / encode them back (mvedit_3d_pipeline.py:1258-1262, :1439-1443) --------
    from mvedit_amd.vae import AutoencoderKLEngine, SD_VAE_CONFIG
    vae = AutoencoderKLEngine.from_state_dict(SY.make_vae_state_dict(dict(SD_VAE_CONFIG), dtype=torch.float16), dict(SD_VAE_CONFIG), torch.float16, dev)
    zs = torch.randn(VIEWS, 4, LATENT, LATENT, device=dev, dtype=torch.float16)
    im = torch.rand(VIEWS, 3, 8 * LATENT, 8 * LATENT, device=dev, dtype=torch.float16) * 2 - 1
    for name, half, inp in (('vae_decode', vae.decoder, zs), ('vae_encode', vae.encoder, im)):
        t = timed(lambda: half.run(inp, 8), it=2)
        fl = sum(half.plan(8, inp.shape[2], inp.shape[3], torch.float16)['flops'].values()) / 8 * VIEWS
        out[name] = dict(ms=round(t * 1e3, 2), views=VIEWS, ms_per_view=round(t * 1e3 / VIEWS, 3), tflops_per_s=round(fl / t / 1e12, 1))
    # ---- SRVGGNetCompact x4 image enhancer on the views rendered at 128^2 / 256^2 (mvedit_3d_pipeline.py:1399-1400; 64 features, 32 convs) ----
    from mvedit_amd.image_enhancer import SRVGGNetCompactEngine
    enh = SRVGGNetCompactEngine(3, 3, 64, 32, 4, dtype=torch.float16, device=dev).load_state_dict(SY.make_srvgg_state_dict(dtype=torch.float16))
    for side in (128, 256):
        lo_res = torch.rand(VIEWS, 3, side, side, device=dev, dtype=torch.float16)
        t = timed(lambda: enh(lo_res), it=2)
        fl = enh.plan(VIEWS, side, side, torch.float16)['flops']['conv']
        out[f'image_enhancer_{side}'] = dict(ms=round(t * 1e3, 2), views=VIEWS, ms_per_view=round(t * 1e3 / VIEWS, 3), tflops_per_s=round(fl / t / 1e12, 1))
    del enh
    # ---- TRACER-B7 foreground masks of the V denoised views (adapter3d_mixin.py:14-19 -> tracer_b7.py:56-73; 640^2 input, bf16; the reference's
    # batch_size = 8 is honoured as a lower bound: the engine walks chunks of 32 views, bitwise the same masks) ----
    from mvedit_amd.segmentor import TracerUniversalB7Engine
    seg = TracerUniversalB7Engine(input_image_size=640, batch_size=8, torch_dtype='bfloat16', erosion=1, device=dev).load_state_dict(SY.make_tracer_state_dict(3))
    views = torch.rand(VIEWS, 3, 8 * LATENT, 8 * LATENT, device=dev)
    t = timed(lambda: seg(views), it=2)
    out['tracer_b7_masks'] = dict(ms=round(t * 1e3, 2), views=VIEWS, ms_per_view=round(t * 1e3 / VIEWS, 3), input=640)
    for k, v in out.items():                      # the MFMA-bound secondary networks against the same dense fp16 peak as the headline
        if isinstance(v, dict) and 'tflops_per_s' in v:
            v['frac_of_mfma_peak'] = round(v['tflops_per_s'] / PEAK_TFLOPS_F16, 4)
    return out


def make_passes(wl, cfg, V, lo, hi, v_loc, world, dev, dtype):
    """Synthetic inputs of one step of workload `wl`, resident in HBM -> (passes, forwards, metric, workload, v_loc, lo, hi); a pass is
    (sample, timesteps, context, num_cross_attn_imgs, cross_attention_kwargs or None)."""
    g = torch.Generator().manual_seed(0)
    cdim = cfg['cross_attention_dim']
    # passes of one step: (sample, timesteps, context, num_cross_attn_imgs, cross_attention_kwargs or None)
    if wl == 'zero123pp':
        # lib/pipelines/zero123plus.py:107-150, :349-350: per denoise step the condition latent (one 320^2 image = 40x40) runs through the UNet
        # writing its self-attention keys / values (ReferenceOnlyAttnProc mode 'w'), then the 3 x 2 tiling of the six views (960 x 640 = a
        # 120 x 80 latent) reads them (mode 'r'); CFG pair, the CFG-first item exempt from the reference.  Not view-sharded: N ranks = N replicas.
        forwards = 2
        ref_dict = {}
        cond = torch.randn(2, 4, 40, 40, generator=g).to(dev, dtype)
        x = torch.randn(2, 4, 120, 80, generator=g).to(dev, dtype)
        ctx = torch.randn(2, CTX_LEN, cdim, generator=g).to(dev, dtype)
        t2 = torch.full((2,), 400.0, device=dev)
        passes = [(cond, t2, ctx, 1, dict(mode='w', ref_dict=ref_dict, is_cfg_guidance=True)),
                  (x, t2, ctx, 1, dict(mode='r', ref_dict=ref_dict, is_cfg_guidance=True))]
        v_loc, lo, hi = 6, 0, 6
        metric = 'Zero123++ denoise-steps/sec (6 views 320^2 tiled to 960x640, SD-2.1, reference-only attention, CFG)'
        workload = ('one Zero123++ denoise step (BASELINE config 2): SD-2.1 UNet on the 40x40 condition latent (writes reference keys / values) + on the '
                    '120x80 latent of the six tiled views (reads them; 9600 + 1600-token self-attention), CFG pair')
    else:
        latents_all = torch.randn(V, 4, LATENT, LATENT, generator=g)
        ctx_uncond = torch.randn(1, CTX_LEN, cdim, generator=g)
        ctx_text = torch.randn(V, CTX_LEN, cdim, generator=g)
        lat = latents_all[lo:hi].to(dev, dtype)
        sample = torch.cat([lat, lat], 0).contiguous()                                   # [uncond | text] halves
        ctx = torch.cat([ctx_uncond.expand(v_loc, -1, -1), ctx_text[lo:hi]], 0).to(dev, dtype).contiguous()
        n_img = 1
        if wl == 'use_reference':
            # adapter3d_mixin.py:86-94: every latent arrives stacked on its reference view's ([b, 4, 128, 64]) and is unrolled to two
            # 64x64 images that share one 2 x 4096-token self-attention (CrossImageAttnProcWrapper, joint_attn.py:11-37)
            ref = torch.randn(V, 4, LATENT, LATENT, generator=g)[lo:hi].to(dev, dtype)
            pair = torch.stack([torch.cat([ref, ref], 0), sample], 1)                    # [2 v, 2, 4, 64, 64]: (reference, view)
            sample = pair.reshape(-1, 4, LATENT, LATENT).contiguous()
            ctx = ctx.unsqueeze(1).expand(-1, 2, -1, -1).reshape(-1, CTX_LEN, cdim).contiguous()
            n_img = 2
        forwards = sample.shape[0]
        passes = [(sample, torch.full((sample.shape[0],), 499.0, device=dev), ctx, n_img, dict(num_cross_attn_imgs=n_img) if n_img > 1 else None)]
        metric = 'multi-view denoise-steps/sec (32 views, 512^2)' + (' with reference-view pairing' if wl == 'use_reference' else '')
        workload = (f'{V}-view 512x512 get_noise_pred: {forwards * world} SD-1.5 UNet forwards (64x64 latents, ctx 77x768) + CFG per step; ControlNet residuals zero'
                    + ('; use_reference: (reference, view) pairs share one 2 x 4096-token self-attention per level-0 block' if wl == 'use_reference' else ''))
    return passes, forwards, metric, workload, v_loc, lo, hi


def measure_workload(dev, wl, dtype, residual_pair, steps=3, warmup=1, parity_ref=None):
    """A compact line for one more workload / mode on this GPU (N = 1): the same step the headline times (make_passes + per-op HIP events), `steps`
    timed steps.  parity_ref = (x, ctx, fp32 output) of a single oracle forward to compare the engine with, or None."""
    from mvedit_amd import ops, synthetic as U
    from mvedit_amd.unet import SD15_CONFIG, SD21_CONFIG, UNet2DConditionEngine
    cfg = dict(SD21_CONFIG if wl == 'zero123pp' else SD15_CONFIG)
    eng = UNet2DConditionEngine.from_state_dict(U.make_state_dict(cfg, seed=1234, dtype=dtype), cfg, dtype, dev)
    eng.set_residual_pair(residual_pair)
    passes, forwards, metric, workload, _, _, _ = make_passes(wl, cfg, VIEWS, 0, VIEWS, VIEWS, 1, dev, dtype)
    optabs = [None] * len(passes)

    def step(profile):
        mss, out = [], None
        for pi, (x_, t_, c_, n_, kw_) in enumerate(passes):
            eng._set_attention(kw_, x_.shape[0], x_.shape[2], x_.shape[3])
            eng.plan(x_.shape[0], x_.shape[2], x_.shape[3], CTX_LEN, n_, False, dtype)
            if profile:
                out, ms = eng._run(0, x_, t_, c_, n_, None, None, None, profile=True)
                if optabs[pi] is None:
                    optabs[pi] = eng.op_table()
                mss.append(ms)
            else:
                out = eng._run(0, x_, t_, c_, n_, None, None, None)
        if wl == 'use_reference':
            out = out.view(-1, 2, *out.shape[1:])[:, 1]
        half = out.shape[0] // 2
        ops.cfg_combine(out[:half].float().contiguous(), out[half:].float().contiguous(), GUIDANCE)
        return mss
    for _ in range(max(warmup, 1)):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    torch.cuda.synchronize()
    ms_per_step = (time.perf_counter() - t0) / steps * 1e3
    per_op = step(True)                                    # one more, untimed, step with per-op events for the class breakdown
    cls_ms, cls_fl = {}, {}
    for optab, ms_list in zip(optabs, per_op):
        for (ph, cls, fl, lab), m in zip(optab, ms_list):
            cls_ms[cls] = cls_ms.get(cls, 0.0) + m
            cls_fl[cls] = cls_fl.get(cls, 0.0) + fl
    gemm_ms = cls_ms.get('conv3x3', 0.0) + cls_ms.get('linear', 0.0)
    gemm_fl = cls_fl.get('conv3x3', 0.0) + cls_fl.get('linear', 0.0)
    dom = 'gemm' if gemm_ms >= cls_ms.get('attention', 0.0) else 'attention'
    ach = (gemm_fl / gemm_ms if dom == 'gemm' else cls_fl['attention'] / cls_ms['attention']) / 1e9
    total_fl = sum(cls_fl.values())
    rec = dict(workload=wl + ('' if residual_pair else ' + plain 16-bit residual stream (the reference\'s rounding points)'), residual_stream='pair' if residual_pair else '16-bit',
               dtype='f16' if dtype == torch.float16 else 'bf16', steps=steps,
               ms_per_step=round(ms_per_step, 3), value=round(1e3 / ms_per_step, 4), unit='denoise-steps/s', forwards_per_step=forwards,
               model_tflops_per_s=round(total_fl / ms_per_step / 1e9, 1),
               roofline=dict(bound='mfma', kernel='k_gemm_pp (conv3x3 + linear)' if dom == 'gemm' else 'k_attention3 / k_attention2', achieved=round(ach, 1),
                             peak=PEAK_TFLOPS_F16, unit='TFLOP/s', frac=round(ach / PEAK_TFLOPS_F16, 4),
                             per_class_ms={k: round(v, 3) for k, v in cls_ms.items()}))
    if parity_ref is not None and isinstance(parity_ref[0], str) and parity_ref[0] == 'zero123pp':
        try:            # the step's own two passes (condition pass writes the reference keys / values, the tiled-view pass reads them) against the oracle's
            out = None
            for (x_, t_, c_, n_, kw_) in passes:
                eng._set_attention(kw_, x_.shape[0], x_.shape[2], x_.shape[3])
                out = eng._run(0, x_, t_, c_, n_, None, None, None)
            got, bout = out.float().cpu(), parity_ref[2]
            rec['parity'] = dict(rel_l2_vs_fp32_oracle=round(float((got - bout).norm() / bout.norm()), 6), shape=list(got.shape), north_star_bar=1e-3,
                                 note='the tiled-view pass of the timed step (reference-only attention over the condition pass\'s keys / values) vs fp32 oracle arithmetic')
        except Exception as e:
            rec['parity'] = {'error': repr(e)[:200]}
    elif parity_ref is not None:
        try:
            bx, bctx, bout, n_img = parity_ref
            eng._set_attention(dict(num_cross_attn_imgs=n_img) if n_img > 1 else None, bx.shape[0], bx.shape[2], bx.shape[3])
            got = eng._run(0, bx.to(dev, dtype), torch.full((bx.shape[0],), 499.0, device=dev), bctx.to(dev, dtype), n_img, None, None, None).float().cpu()
            rec['parity'] = dict(rel_l2_vs_fp32_oracle=round(float((got - bout).norm() / bout.norm()), 6), shape=list(bx.shape), north_star_bar=1e-3)
        except Exception as e:
            rec['parity'] = {'error': repr(e)[:200]}
    del eng
    torch.cuda.empty_cache()
    return rec


def outer_step(dev, n_optim_timed=24):
    """One iteration of the reference's outer loop at V = 32 (lib/pipelines/mvedit_3d_pipeline.py:1141-1479; defaults of lib/core/webui/parameters.py and
    tab_3d_to_3d.py:14: diff_bs 6, render_bs 6, patch_size 128, patch_bs_nerf 1, patch_bs 8, n_inverse_steps 96), composed from the engine's
    own objects the way `__call__` composes the reference's, each stage timed with HIP events on the launch stream:
      noise   : Adapter3DMixin.get_noise_pred over all 2 V images with TWO ControlNets (tile + depth; :1246-1249 -> adapter3d_mixin.py:68-135)
      decode  : x0 prediction, vae.decode of the V latents, (x / 2 + 0.5).clamp, NHWC (:1252-1263)
      masks   : get_tgt_masks = TRACER-B7 at 640^2 (:1266)
      optim   : `n_inverse_steps` iterations of nerf_optim (:507-633; 128^2 rays: march -> decode -> composite -> losses + LPIPS patch -> backward -> Adam),
                or of mesh_optim once DMTet has taken over (:716-847; render_bs views at 512^2, patch_bs LPIPS patches, regularisers) -- both timed
      render  : BaseNeRF.render of the V views in render_bs batches + shading / tone mapping (:1341-1389)
      encode  : vae.encode of the V rendered views (:1439-1443)
    Synthetic scene and weights; the figures are per-stage wall times, not a quality run.  Everything but `optim` is per view (shards with the
    views); `optim` is the replicated 3D update every rank repeats."""
    import math
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from scene import icosphere, sphere_density_grid
    from mvedit_amd import nerf, raymarching as rm, synthetic as SY
    from mvedit_amd.controlnet import ControlNetEngine, MultiControlNetEngine
    from mvedit_amd.lpips import LPIPSEngine
    from mvedit_amd.mesh_ops import Mesh, MeshRenderer, mesh_regularizers
    from mvedit_amd.pipelines import Adapter3DMixin
    from mvedit_amd.pipelines.diffusion import predict_x0
    from mvedit_amd.recon_loss import mesh_optim_loss, nerf_optim_loss
    from mvedit_amd.segmentor import TracerUniversalB7Engine
    from mvedit_amd.tonemapping import Tonemapping, make_shading_fun, shade_views
    from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine
    from mvedit_amd.vae import AutoencoderKLEngine, SD_VAE_CONFIG

    V, S, f16 = VIEWS, 8 * LATENT, torch.float16
    g = torch.Generator().manual_seed(11)

    def timed(fn, it=2, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            out = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it, out

    out = {}
    # ---- the networks of the step ----------------------------------------------------------------------------------------------------
    class Pipe(Adapter3DMixin):
        pass
    pipe = Pipe()
    cfg = dict(SD15_CONFIG)
    pipe.unet = UNet2DConditionEngine.from_state_dict(SY.make_state_dict(cfg, seed=1234, dtype=f16), cfg, f16, dev)
    cn_sd = SY.make_controlnet_state_dict(cfg, dtype=f16)
    pipe.controlnet = MultiControlNetEngine([ControlNetEngine.from_state_dict(cn_sd, cfg, f16, dev) for _ in range(2)])
    del cn_sd
    pipe.segmentation = TracerUniversalB7Engine(input_image_size=640, batch_size=8, torch_dtype='bfloat16', erosion=1, device=dev).load_state_dict(SY.make_tracer_state_dict(3))
    pipe.bg_color = 1.0
    vae = AutoencoderKLEngine.from_state_dict(SY.make_vae_state_dict(dict(SD_VAE_CONFIG), dtype=f16), dict(SD_VAE_CONFIG), f16, dev)
    lat = torch.randn(V, 4, LATENT, LATENT, generator=g).to(dev, f16)
    ctx = torch.randn(2 * V, CTX_LEN, cfg['cross_attention_dim'], generator=g).to(dev, f16)
    ctrl_img = torch.rand(V, 3, S, S, generator=g).to(dev, f16)
    ctrl_dep = torch.rand(V, 3, S, S, generator=g).to(dev, f16)
    two = lambda x: torch.cat([x, x], 0)
    t_step = torch.full((2 * V,), 499.0, device=dev)
    # the reference's call, exactly as its ordinary (not use_reference) 1-pass branch builds the lists (mvedit_3d_pipeline.py:1236-1249, diff_bs 6):
    # `torch.cat([x] * 2).split(diff_bs)` for the latents AND the control images -- 11 chunks of <= 6, the sixth straddling the CFG halves.  The
    # mixin fuses the chunks into one 64-image launch; Adapter3DMixin._cat_shared_cond views the control chunks as one tensor again, finds its
    # halves equal (one device comparison) and hands over one half: the ControlNet engines embed each control image once for both halves.
    # (round 4 timed `x.split(diff_bs) * 2` control lists against same-shape latents, a combination the reference never builds.)
    DIFF_BS = 6
    mk = lambda: (list(two(lat).split(DIFF_BS)), list(ctx.split(DIFF_BS)), list(two(ctrl_img).split(DIFF_BS)), list(two(ctrl_dep).split(DIFF_BS)))
    lists = mk()
    assert len({len(x) for x in lists}) == 1
    ms, noise = timed(lambda: pipe.get_noise_pred(*lists, t_step, 1.0, 1.0, GUIDANCE))
    out['noise_pred_unet_2_controlnets_ms'] = round(ms, 2)
    pipe.detect_repeated_cond = False                                      # the same call without the recognition: every control image embedded twice
    ms2, noise2 = timed(lambda: pipe.get_noise_pred(*lists, t_step, 1.0, 1.0, GUIDANCE))
    pipe.detect_repeated_cond = True
    out['noise_pred_unet_2_controlnets_unshared_ms'] = round(ms2, 2)
    out['noise_pred_shared_equals_unshared'] = bool(torch.equal(noise, noise2))
    del noise2, lists

    def decode():
        x0 = predict_x0(lat, noise, 0.6, 0.8) / 0.18215
        img = torch.cat([vae.decode(x0[i:i + 8].to(f16), return_dict=False)[0] for i in range(0, V, 8)])
        return (img / 2 + 0.5).clamp(min=0, max=1).permute(0, 2, 3, 1)[None].to(torch.float32)
    ms, tgt_images = timed(decode)
    out['x0_vae_decode_ms'] = round(ms, 2)
    ms, tgt_masks = timed(lambda: pipe.get_tgt_masks(tgt_images, 0))
    out['tracer_masks_ms'] = round(ms, 2)
    ms, _ = timed(lambda: torch.cat([vae.encode(tgt_images[0, i:i + 8].permute(0, 3, 1, 2).to(f16) * 2 - 1, return_dict=False)[0].mean for i in range(0, V, 8)]))
    out['vae_encode_ms'] = round(ms, 2)
    del pipe, vae, noise
    torch.cuda.empty_cache()

    # ---- NeRF side: one nerf_optim iteration (128^2 rays of one view) and the render of the V views ----------------------------------
    meta, rows = nerf.grid_meta(12, 16, 320)
    table = (torch.rand(rows, 2, generator=g) * 2 - 1) * 0.1
    w1 = (torch.rand(64, 24, generator=g) * 2 - 1) * math.sqrt(6 / (64 + 24))
    w2 = (torch.rand(4, 64, generator=g) * 2 - 1) * math.sqrt(6 / (4 + 64))
    dec = nerf.INGPDecoderParams(table, w1, torch.zeros(64), w2, torch.tensor([2.0, 0.0, 0.0, 0.0]), 12, 320, device=dev)
    bits = rm.packbits(torch.from_numpy(sphere_density_grid(128, radius=0.5)).to(dev), 0.5)
    fl = S / (2 * math.tan(math.radians(15)))
    intr = torch.tensor([[fl, fl, S / 2, S / 2]] * V, device=dev)
    poses = surround_poses(V).to(dev)
    tm = Tonemapping(device=dev)
    lights = torch.nn.functional.normalize(torch.tensor([[0.3, -0.5, -1.0]] * V, device=dev), dim=-1)
    nr = nerf.NeRFRenderer(grid_size=128)
    rcfg = dict(return_rgba=True, compute_normal=True, dt_gamma_scale=0.0)

    def render():
        imgs = []
        for i in range(0, V, 6):                                           # render_bs = 6
            rgba, depth, normal, normal_fg = nr.render(dec, None, bits[None], S, S, intr[None, i:i + 6], poses[None, i:i + 6], cfg=rcfg)
            imgs.append(shade_views(rgba, normal_fg, lights[i:i + 6], 0.2, 1.0, tm))
        return imgs
    with torch.no_grad():
        ms, _ = timed(render)
    out['nerf_render_32_views_512_ms'] = round(ms, 2)

    ps, P = 128, 1                                                         # patch_size, patch_bs_nerf: n_inverse_rays = 128^2
    ro, rd, _ = nerf.camera_rays(intr[:P] * (ps / S), poses[:P], ps, ps)   # one whole view at patch resolution stands in for the sampled patch
    ys, xs = torch.meshgrid(torch.arange(ps, dtype=torch.float32), torch.arange(ps, dtype=torch.float32), indexing='ij')
    flp = ps / (2 * math.tan(math.radians(15)))
    dirs = torch.stack([(xs + 0.5 - ps / 2) / flp, (ys + 0.5 - ps / 2) / flp, torch.ones_like(xs)], -1)[None].repeat(P, 1, 1, 1).to(dev)
    for t in dec.parameters().values():
        t.requires_grad_(True)
    dec.max_steps = 512
    vr = nerf.VolumeRenderer(dec)
    vr.training = True
    tgt_m = torch.rand(P, ps, ps, 1, generator=g).to(dev)
    tgt_rgb = torch.rand(P, ps, ps, 3, generator=g).to(dev)
    lp = LPIPSEngine.from_state_dict({k: v.to(torch.bfloat16).float() for k, v in SY.make_lpips_state_dict().items()}, torch.bfloat16, device=dev)
    opt = torch.optim.Adam(list(dec.parameters().values()), lr=1e-2, eps=1e-15)

    def nerf_iter():
        opt.zero_grad()
        o = vr.forward(ro, rd, bits, 128, dt_gamma=0.0)
        res = nerf_optim_loss(o['image'], o['weights_sum'], o['depth'], o['weights'], o['ts'][0], tgt_rgb, tgt_m, dirs, torch.ones(P, device=dev),
                              lights[:P], tonemapping=tm, shaded=True, normal_reg_weight=0.5, entropy_weight=0.2)
        loss = res['loss']
        if lp is not None:
            loss = loss + 0.3 * lp(res['out_rgbs'].permute(0, 3, 1, 2), tgt_rgb.permute(0, 3, 1, 2)).mean()
        loss.backward()
        opt.step()
    ms, _ = timed(nerf_iter, it=n_optim_timed, warm=4)
    out['nerf_optim_iter_ms'] = round(ms, 3)
    if lp is not None:
        a = torch.rand(8, 3, 128, 128, device=dev, requires_grad=True)
        b = torch.rand(8, 3, 128, 128, device=dev)

        def lp_fb():
            a.grad = None
            lp(a, b).mean().backward()
        ms, _ = timed(lp_fb, it=3)
        out['lpips_ms'] = round(ms, 3)                                      # 8 patches of 128^2, forward + backward w.r.t. the prediction (patch_bs = 8)

    # ---- mesh side: one mesh_optim iteration (render_bs = 6 views at 512^2, 8 LPIPS patches, regularisers, Adam on the vertices) --------
    v0, fc = icosphere(6, 0.6)
    faces = torch.from_numpy(fc).to(dev)
    mr = MeshRenderer(near=0.01, far=100)
    nvm = 6
    shade = make_shading_fun(lights[:nvm, None, None, :].expand(nvm, S, S, 3).contiguous(), 0.2, tm)
    verts = torch.from_numpy(v0).to(dev).requires_grad_(True)
    mopt = torch.optim.Adam([verts], lr=1e-3)
    tgt_rgb6 = torch.rand(nvm, S, S, 3, generator=g).to(dev)
    tgt_m6 = (torch.rand(nvm, S, S, 1, generator=g) > 0.5).float().to(dev)
    erode = -torch.nn.functional.max_pool2d(-tgt_m6.permute(0, 3, 1, 2), 5, stride=1, padding=2).permute(0, 2, 3, 1).contiguous()
    ysf, xsf = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing='ij')
    dirs6 = torch.stack([(xsf + 0.5 - S / 2) / fl, (ysf + 0.5 - S / 2) / fl, torch.ones_like(xsf)], -1)[None].repeat(nvm, 1, 1, 1).to(dev)
    normal_t = torch.rand(nvm, S, S, 3, generator=g).to(dev)

    def mesh_iter():
        mopt.zero_grad()
        m = Mesh(verts, faces, vc=torch.cat([torch.full_like(verts, 0.7), torch.ones_like(verts[:, :1])], -1))
        m.auto_normal()
        o = mr([m], poses[None, :nvm], intr[None, :nvm], S, S, shading_fun=shade, normal_bg=[0.5, 0.5, 1.0])
        res = mesh_optim_loss(o['rgba'][0], o['normal'][0], o['depth'][0].detach(), tgt_rgb6, erode, tgt_m6, dirs6, torch.ones(nvm, device=dev),
                              target_n=normal_t, normal_reg_weight=1.0)
        lap, nc = mesh_regularizers(verts, faces, m.face_normals)
        loss = res['loss'] + 5.0 * (lap + nc)
        if lp is not None:                                                   # patch_bs = 8 patches of 128^2 cut from the rendered views
            cut = lambda x: torch.stack([x[i % nvm, 128 * (i // nvm):128 * (i // nvm) + 128, 64:192] for i in range(8)]).permute(0, 3, 1, 2)
            loss = loss + 0.3 * lp(cut(res['out_rgbs']), cut(tgt_rgb6)).mean()
        loss.backward()
        mopt.step()
    ms, _ = timed(mesh_iter, it=n_optim_timed, warm=4)
    out['mesh_optim_iter_ms'] = round(ms, 3)

    try:
        out['texture_superres'] = texture_superres(dev, lp)
    except Exception as e:      # a composed figure must not take the others with it
        out['texture_superres'] = {'error': repr(e)[:300]}

    n_inv = 96
    per_view = out['noise_pred_unet_2_controlnets_ms'] + out['x0_vae_decode_ms'] + out['tracer_masks_ms'] + out['nerf_render_32_views_512_ms'] + out['vae_encode_ms']
    rep = n_inv * out['nerf_optim_iter_ms']
    out['outer_step_ms'] = dict(n_inverse_steps=n_inv, sharded_per_view_stages=round(per_view, 1), replicated_nerf_optim=round(rep, 1),
                                total_nerf_stage=round(per_view + rep, 1), total_mesh_stage=round(per_view + n_inv * out['mesh_optim_iter_ms'], 1),
                                amdahl_speedup_at_8_gpus=round((per_view + rep) / (per_view / 8 + rep), 2),
                                note='sharded stages divide by the rank count (views partition); the optimiser iterations are replicated on every rank')
    return out


def texture_superres(dev, lp):
    """BASELINE config 4's second half composed end to end: the texture super-resolution loop of the 3D-to-3D pipelines
    (lib/pipelines/mvedit_texture_superres_pipeline.py:171-470 as lib/apis/adapter3d.py:578-620 calls it with the web UI defaults of
    lib/core/webui/shared_opts.py:245-275: 6 views + 2 regularisation views at 512^2, 24 steps at denoising strength 0.4 = 10 denoise steps,
    two ControlNets (tile + depth), n_inverse_steps 48, patch_size 512, patch_bs 1, render_bs 8, 2048^2 atlas):
      encode  : vae.encode of the 6 rendered views (:323-327)
      denoise : get_noise_pred over the 12 images of a step (the reference's `torch.cat([x] * 2).split(diff_bs)` lists, :377-389) x 10 steps
      decode  : x0 prediction + vae.decode of the 6 latents at the last step (:391-402)
      optim   : 48 x texture_optim iterations (:89-163): MeshRenderer.forward of 8 views at 512^2 with the hash-grid decoder as the albedo
                (make_nerf_albedo_shading_fun), weighted L1 x 2 + LPIPS on one 512^2 patch, backward through antialias / interpolate /
                the decoder, Adam on the decoder
      bake    : bake_xyz_shading_fun of the decoder into a 2048^2 atlas + edge dilation (:462-466)
    Synthetic mesh / weights; per-stage wall times with HIP events.  Every stage but `optim` and `bake` is per view."""
    import math
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from scene import face_atlas, icosphere
    from mvedit_amd import nerf, synthetic as SY
    from mvedit_amd.controlnet import ControlNetEngine, MultiControlNetEngine
    from mvedit_amd.mesh_ops import Mesh, MeshRenderer
    from mvedit_amd.pipelines import Adapter3DMixin
    from mvedit_amd.pipelines.diffusion import predict_x0
    from mvedit_amd.tonemapping import make_nerf_albedo_shading_fun
    from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine
    from mvedit_amd.vae import AutoencoderKLEngine, SD_VAE_CONFIG

    V, NREG, S, f16 = 6, 2, 8 * LATENT, torch.float16
    STEPS, N_INV, DIFF_BS, RENDER_BS, MAP = 10, 48, 6, 8, 2048
    g = torch.Generator().manual_seed(23)

    def timed(fn, it=2, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            o = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it, o

    out = {}

    class Pipe(Adapter3DMixin):
        pass
    pipe = Pipe()
    cfg = dict(SD15_CONFIG)
    pipe.unet = UNet2DConditionEngine.from_state_dict(SY.make_state_dict(cfg, seed=1234, dtype=f16), cfg, f16, dev)
    cn_sd = SY.make_controlnet_state_dict(cfg, dtype=f16)
    pipe.controlnet = MultiControlNetEngine([ControlNetEngine.from_state_dict(cn_sd, cfg, f16, dev) for _ in range(2)])
    del cn_sd
    vae = AutoencoderKLEngine.from_state_dict(SY.make_vae_state_dict(dict(SD_VAE_CONFIG), dtype=f16), dict(SD_VAE_CONFIG), f16, dev)
    views = torch.rand(V, 3, S, S, generator=g).to(dev, f16)
    ms, lat = timed(lambda: vae.encode(views * 2 - 1, return_dict=False)[0].mean * 0.18215)
    out['vae_encode_6_views_ms'] = round(ms, 2)
    lat = lat.to(f16)
    ctx = torch.randn(2 * V, CTX_LEN, cfg['cross_attention_dim'], generator=g).to(dev, f16)
    ctrl_dep = torch.rand(V, 3, S, S, generator=g).to(dev, f16)
    two = lambda x: torch.cat([x, x], 0)
    t_step = torch.full((2 * V,), 399.0, device=dev)
    lists = (list(two(lat).split(DIFF_BS)), list(ctx.split(DIFF_BS)), list(two(views).split(DIFF_BS)), list(two(ctrl_dep).split(DIFF_BS)))
    ms, noise = timed(lambda: pipe.get_noise_pred(*lists, t_step, 1.0, 1.0, GUIDANCE), it=3)
    out['noise_pred_12_images_2_controlnets_ms'] = round(ms, 2)

    def decode():
        x0 = predict_x0(lat, noise, 0.6, 0.8) / 0.18215
        img = vae.decode(x0.to(f16), return_dict=False)[0]
        return (img / 2 + 0.5).clamp(min=0, max=1).permute(0, 2, 3, 1).to(torch.float32)
    ms, tgt = timed(decode)
    out['x0_vae_decode_6_views_ms'] = round(ms, 2)
    del pipe, vae, noise, lists
    torch.cuda.empty_cache()

    # ---- texture field + mesh ----------------------------------------------------------------------------------------------------------
    meta, rows = nerf.grid_meta(12, 16, 320)
    table = (torch.rand(rows, 2, generator=g) * 2 - 1) * 0.1
    w1 = (torch.rand(64, 24, generator=g) * 2 - 1) * math.sqrt(6 / (64 + 24))
    w2 = (torch.rand(4, 64, generator=g) * 2 - 1) * math.sqrt(6 / (4 + 64))
    dec = nerf.INGPDecoderParams(table, w1, torch.zeros(64), w2, torch.tensor([2.0, 0.0, 0.0, 0.0]), 12, 320, device=dev)
    for t in dec.parameters().values():
        t.requires_grad_(True)
    v0, fc = icosphere(6, 0.6)
    vt, ft = face_atlas(fc)
    tv = lambda a: torch.from_numpy(a).to(dev)
    mesh = Mesh(tv(v0), tv(fc), tv(vt), tv(ft), vn=tv((v0 / np.linalg.norm(v0, axis=-1, keepdims=True)).astype(np.float32)), fn=tv(fc))
    mr = MeshRenderer(near=0.01, far=100)
    fl = S / (2 * math.tan(math.radians(20)))
    NC = V + NREG
    intr = torch.tensor([[fl, fl, S / 2, S / 2]] * NC, device=dev)
    poses = surround_poses(NC, radius=3.1).to(dev)
    tgt_all = torch.cat([tgt, torch.rand(NREG, S, S, 3, generator=g).to(dev)], 0)
    wts = torch.rand(NC, S, S, 1, generator=g).to(dev)
    shade = make_nerf_albedo_shading_fun(lambda p: dec.point_decode_autograd(p)[1])
    opt = torch.optim.Adam(list(dec.parameters().values()), lr=1e-2, eps=1e-15)
    bg = 0.5

    def tex_iter():
        opt.zero_grad()
        ib = intr.clone()
        ib[:, 2:] += (torch.rand_like(ib[:, 2:]) - 0.5) / mr.ssaa
        o = mr([mesh], poses[None, :RENDER_BS], ib[None, :RENDER_BS], S, S, shade)
        rgba = o['rgba'][0]
        rgb = rgba[..., :3] + (1 - rgba[..., 3:].clamp(min=1e-3)) * bg
        loss = ((rgb - tgt_all[:RENDER_BS]).abs() * wts[:RENDER_BS]).mean() * 2                 # nerf.pixel_loss = weighted L1LossMod (texture_optim :128-131)
        if lp is not None:
            w_p = wts[0].amax()
            loss = loss + 0.3 * w_p * lp(rgb[:1].permute(0, 3, 1, 2), tgt_all[:1].permute(0, 3, 1, 2)).mean()   # one 512^2 patch (patch_bs 1)
        loss.backward()
        opt.step()
        return loss
    ms, _ = timed(tex_iter, it=12, warm=3)
    out['texture_optim_iter_ms'] = round(ms, 3)

    def bake():
        with torch.no_grad():
            return mr.bake_xyz_shading_fun([mesh], make_nerf_albedo_shading_fun(lambda p: dec.point_decode(p)[1]), map_size=MAP)
    ms, baked = timed(bake, it=2)
    out['bake_xyz_2048_ms'] = round(ms, 2)
    assert baked[0].albedo.shape == (MAP, MAP, 4)
    per_view = out['vae_encode_6_views_ms'] + STEPS * out['noise_pred_12_images_2_controlnets_ms'] + out['x0_vae_decode_6_views_ms']
    rep = N_INV * out['texture_optim_iter_ms'] + out['bake_xyz_2048_ms']
    out['total_ms'] = dict(denoise_steps=STEPS, n_inverse_steps=N_INV, per_view_stages=round(per_view, 1), texture_optim_and_bake=round(rep, 1),
                           total=round(per_view + rep, 1),
                           note='6 views + 2 regularisation views, 512^2, 2048^2 atlas; per-view stages shard with the views, the texture optimisation and the '
                                'bake are replicated')
    return out


def scaling_projection(eng, dev, dtype, ms_full):
    """The forward a rank of an N-GPU job runs (64 / N images), timed on THIS GPU with the headline engine, and what strong scaling that projects
    before / after the step's all-gather (128 MiB of maps over 7 xGMI links: ~0.4 ms assumed, measured by `collectives` when N > 1)."""
    rows = {}
    g = torch.Generator().manual_seed(1)
    for n in (2, 4, 8):
        B = 2 * VIEWS // n
        x = torch.randn(B, 4, LATENT, LATENT, generator=g).to(dev, dtype)
        c = torch.randn(B, CTX_LEN, 768, generator=g).to(dev, dtype)
        t = torch.full((B,), 499.0, device=dev)
        eng._set_attention(None, B, LATENT, LATENT)
        for _ in range(2):
            eng._run(0, x, t, c, 1, None, None, None)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(4):
                eng._run(0, x, t, c, 1, None, None, None)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 4 * 1e3)
        rows[str(n)] = dict(images_per_rank=B, ms=round(best, 3), speedup_before_collectives=round(ms_full / best, 2),
                            speedup_with_all_gather=round(ms_full / (best + 0.4), 2))
    return dict(per_n_gpus=rows, all_gather_ms_assumed=0.4, base_ms=round(ms_full, 3),
                note='one rank\'s share of the 32-view step on this box (tools/scale_preview.py); north_star target at N = 8: >= 6x')


if __name__ == '__main__':          # python tools/bench_parts.py texture_superres | outer_step | secondary : one part on its own (development aid)
    what = sys.argv[1] if len(sys.argv) > 1 else 'texture_superres'
    dev_ = torch.device('cuda', 0)
    torch.cuda.set_device(dev_)
    if what == 'texture_superres':
        from mvedit_amd import synthetic as SY_
        from mvedit_amd.lpips import LPIPSEngine as LP_
        lp_ = LP_.from_state_dict({k: v.to(torch.bfloat16).float() for k, v in SY_.make_lpips_state_dict().items()}, torch.bfloat16, device=dev_)
        print(json.dumps(texture_superres(dev_, lp_)))
    elif what == 'outer_step':
        print(json.dumps(outer_step(dev_)))
    else:
        print(json.dumps(secondary(dev_)))
