#!/usr/bin/env python
"""Headline benchmark: multi-view denoise-steps/sec (32 views, 512^2) on N MI355X.

One "step" = one pass of the reference's `Adapter3DMixin.get_noise_pred`
(lib/pipelines/adapter3d_mixin.py:68-135) over ALL V=32 views with classifier-free guidance:
2*V = 64 SD-1.5 UNet forwards on 64x64 latents (512^2 images) followed by the CFG combine.  ControlNet
residuals are zero-valued inputs (UNet-only metric); weights are seeded random
SD-1.5-topology tensors (no checkpoint is reachable offline), data is synthetic and resident in HBM before
the timed region.

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU): the V views are partitioned across
ranks (strong scaling: total work fixed at 32 views); every rank denoises its V/N views, then ONE RCCL
all-gather over xGMI gives every rank the full [V,4,64,64] noise prediction (what the replicated 3D update
consumes).  value = steps/s of the whole 32-view job = 1 / max-over-ranks(step time).

The JSON line also carries
  roofline     : the dominant kernel (k_gemm_pp: 3x3 implicit-GEMM conv + linear launches, MFMA bound): algorithmic FLOPs of all its
                 launches in one step / their summed HIP-event durations, recorded on the launch stream over the K timed steps REPEATED right
                 after the timed region (round 5: the events themselves cost ~2 ms of a 64 ms step, so the timed K steps carry none;
                 `op_timing` has both wall times);
  cpu_baseline : the torch-fp32 oracle of the same UNet timed on this box's host cores on a bounded sample (a forward of two images),
                 scaled to 64 forwards/step.  Baseline only;
  parity       : rows of the LAST TIMED step's 64-image output against fp32 oracle forwards of the same items (north_star: 1e-3).
The engine runs in its default mode: residual stream as an unrounded pair (16-bit value + 8-bit low half), the mode inside north_star's 1e-3;
`--plain-stream` times the reference's 16-bit stream (also reported as extra_workloads[0]).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.bench_parts import (CTX_LEN, GUIDANCE, LATENT, PEAK_TFLOPS_F16, VIEWS, PowerSampler, clock_probe, make_passes,  # noqa: E402
                               measure_workload, outer_step, pmc_traffic, scaling_projection, secondary)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--dtype', default='fp16', choices=['fp16', 'bf16'])
    ap.add_argument('--views', type=int, default=VIEWS)
    ap.add_argument('--workload', default='mvedit32', choices=['mvedit32', 'use_reference', 'zero123pp'],
                    help='mvedit32: the headline (BASELINE configs 3/4: 2V SD-1.5 forwards on 64x64 latents); use_reference: the same step with the '
                         'reference-view pairing of adapter3d_mixin.py:86-94 ([b, 4, 128, 64] latents = 4V forwards, self-attention over 2 x 4096 '
                         'tokens); zero123pp: BASELINE config 2, one Zero123++ denoise step (SD-2.1 on the 120x80 latent of six 320^2 views, '
                         'reference-only attention written by a 40x40 condition pass, CFG pair)')
    ap.add_argument('--plain-stream', action='store_true', help='time the headline with the 16-bit residual stream of the reference\'s half modules '
                    '(UNet2DConditionEngine.set_residual_pair(False): 1.2e-3 from fp32 arithmetic, outside north_star\'s 1e-3).  The DEFAULT since round 5 is '
                    'the engine\'s default mode: the stream as an unrounded (hi, lo) pair, 8.7e-4; the plain mode is reported as an extra workload')
    ap.add_argument('--residual-pair', action='store_true', help='(accepted for older command lines: the pair is the default now)')
    ap.add_argument('--no-extra', action='store_true', help='skip the compact extra workloads (use_reference, zero123pp, bf16, residual pair) and the outer-step figures')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the NeRF / raster / back-projection figures')
    ap.add_argument('--no-op-timing', action='store_true', help='time the steps without per-op HIP events')
    ap.add_argument('--secondary-only', action='store_true', help='run only the render / reconstruct / VAE figures (the workload of the '
                    'rocprofv3 passes behind profiles/rNN_rocprof_render_*.txt) and print them')
    return ap.parse_args()


def cpu_baseline(cfg, dtype=torch.float16, latent_hw=(LATENT, LATENT), B=1, n_img=1, repeats=3, x=None, ctx=None, t=499, zero123pp_passes=None):
    """Oracle (kind "port") on the host cores: a forward of B images, fp32 arithmetic over the engine's (16-bit rounded) weights.
    Returns the baseline record and (x, ctx, out) so that the same forward can be compared with the HIP engine (the oracle as checker).
    The ONLY function of this file that touches oracle/ (tests/test_abi.py).  x / ctx given: those rows (the headline passes rows of the TIMED
    batch, so that the checker sees the launch decisions the timed batch takes); else seeded random inputs.  B / n_img / repeats = 1: the checker
    leg of the extra workloads (one forward of a CFG pair, or of a (reference, view) pair under cross-image attention), not a baseline figure."""
    from oracle import unet_oracle as U
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    sd = {k: v.to(dtype).float() for k, v in U.make_state_dict(cfg, seed=1234).items()}
    if zero123pp_passes is not None:
        # checker leg of BASELINE config 2: the two passes of a Zero123++ denoise step exactly as tools/bench_parts.make_passes builds them -- the condition
        # latent writes the reference keys / values, the tiled views read them, the CFG-first item exempt (zero123plus.py:43-77, :107-150)
        (cond, t2, cctx, _, _), (xx, _, _, _, _) = zero123pp_passes
        f = lambda a: a.detach().float().cpu()
        d = {}
        with torch.no_grad():
            U.unet_forward(sd, cfg, f(cond), f(t2), f(cctx), attn_opts=dict(mode='w', ref_dict=d, ref_skip=1))
            out = U.unet_forward(sd, cfg, f(xx), f(t2), f(cctx), attn_opts=dict(mode='r', ref_dict=d, ref_skip=1))
        return None, ('zero123pp', None, out)
    if x is None:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, 4, *latent_hw, generator=g).to(dtype).float()
        ctx = torch.randn(B, CTX_LEN, cfg['cross_attention_dim'], generator=g).to(dtype).float()
    else:
        x, ctx = x.detach().float().cpu(), ctx.detach().float().cpu()
        B, latent_hw = x.shape[0], tuple(x.shape[2:])
    dts = []
    with torch.no_grad():
        for _ in range(repeats):                 # ~10 s of host work in total: a bounded sample, not the full step
            t0 = time.perf_counter()
            out = U.unet_forward(sd, cfg, x, t, ctx, n_img)
            dts.append(time.perf_counter() - t0)
    dt = min(dts) / B
    rec = dict(value=None, unit='denoise-steps/s', cores=threads, kind='port', seconds_per_forward=round(dt, 3),
               sample=f'{B} UNet forward(s) in one call ({latent_hw[0]}x{latent_hw[1]} latent, fp32 torch oracle), best of {repeats}: {dt:.2f} s per forward; '
                      'scaled linearly to the forwards of a step')
    return rec, (x, ctx, out)


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # MVE_BENCH_FORCE_DIST=1 runs the RCCL code path (init, barrier, all-gather, all-reduce) with a single rank, so that it can
    # be exercised on a 1-GPU box under `torch.distributed.run --nproc-per-node 1`
    use_dist = world > 1 or os.environ.get('MVE_BENCH_FORCE_DIST') == '1'
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback exists for the HIP path)'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if args.secondary_only:
        print(json.dumps({'secondary': secondary(dev)}), flush=True)
        return

    from mvedit_amd.unet import UNet2DConditionEngine, SD15_CONFIG, SD21_CONFIG
    from mvedit_amd import ops
    from mvedit_amd.parallel import partition_views
    from mvedit_amd import synthetic as U   # seeded random weights in diffusers layout (the oracle is used by cpu_baseline only)

    dtype = torch.float16 if args.dtype == 'fp16' else torch.bfloat16
    wl = args.workload
    cfg = dict(SD21_CONFIG if wl == 'zero123pp' else SD15_CONFIG)
    V = args.views
    lo, hi = partition_views(V, world, rank)
    v_loc = hi - lo

    # ---- weights + synthetic inputs, resident in HBM -------------------------------------------------------
    sd = U.make_state_dict(cfg, seed=1234, dtype=dtype)
    eng = UNet2DConditionEngine.from_state_dict(sd, cfg, dtype, dev)
    del sd
    pair_mode = not args.plain_stream
    eng.set_residual_pair(pair_mode)                      # (the engine's default mode; --plain-stream: the reference's rounding points)
    if os.environ.get('MVE_BENCH_GRAPH') == '1':          # experiment: hipGraph replay of the forward (off by default)
        eng.enable_graph(True)
        side = torch.cuda.Stream(dev)                      # stream capture is not allowed on the legacy default stream
        side.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(side)
    passes, forwards, metric, workload, v_loc, lo, hi = make_passes(wl, cfg, V, lo, hi, v_loc, world, dev, dtype)
    infos = [None] * len(passes)      # filled by the first (warm-up) step: the plan depends on the attention mode set per pass
    # The one collective of a pipeline step (BASELINE north_star): every rank contributes the rendered / decoded
    # RGB + alpha + depth + normal maps of ITS views (8 channels x 512^2 fp16 = 4 MiB per view) and receives all V of them before
    # the replicated 3D update.  Synthetic payload of exactly that size; the per-view noise prediction itself stays local.
    shards = use_dist and wl != 'zero123pp'
    maps_local = torch.zeros(v_loc, 8, 8 * LATENT, 8 * LATENT, device=dev, dtype=torch.float16) if shards else None
    gathered = torch.empty(V, 8, 8 * LATENT, 8 * LATENT, device=dev, dtype=torch.float16) if shards else None
    optabs = [None] * len(passes)
    last_raw = [None]                 # the UNet output of the most recent step, all 2 v_loc rows (the checker leg compares rows of the TIMED batch)

    def step(profile):
        """-> (noise prediction, [per-op milliseconds of every pass] or None)"""
        mss = []
        out = None
        for pi, (x_, t_, c_, n_, kw_) in enumerate(passes):
            eng._set_attention(kw_, x_.shape[0], x_.shape[2], x_.shape[3])
            if infos[pi] is None:
                infos[pi] = eng.plan(x_.shape[0], x_.shape[2], x_.shape[3], CTX_LEN, n_, False, dtype)
            if profile:
                out, ms = eng._run(0, x_, t_, c_, n_, None, None, None, profile=True)
                if optabs[pi] is None:
                    optabs[pi] = eng.op_table()
                mss.append(ms)
            else:
                out = eng._run(0, x_, t_, c_, n_, None, None, None)
        last_raw[0] = out
        if wl == 'use_reference':
            out = out.view(-1, 2, *out.shape[1:])[:, 1]                                  # the view's half of every pair
        half = out.shape[0] // 2
        noise = ops.cfg_combine(out[:half].float().contiguous(), out[half:].float().contiguous(), GUIDANCE)
        if shards:
            dist.all_gather_into_tensor(gathered, maps_local)
        return noise, (mss if profile else None)

    for _ in range(max(args.warmup, 1)):
        step(False)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # The timed region: EXACTLY K steps, nothing but the workload (no per-op events: round 5 measured them at 2 ms of a 64 ms step -- 712 event records
    # per step keep kernel tails from overlapping the next launch).
    for i in range(args.steps):
        step(False)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed_rows = None
    if rank == 0 and wl == 'mvedit32' and last_raw[0] is not None:
        # rows of the LAST TIMED step's batch for the checker leg: view 0's unconditional and text rows (rows 0 and v_loc of [uncond | text])
        timed_rows = ([0, v_loc], last_raw[0][[0, v_loc]].float().cpu())
    # The roofline's kernel durations: HIP events around every launch (mve_unet_forward's op_ms, recorded on the launch stream) over the SAME K steps
    # repeated right after the timed ones -- same inputs, same plan, the chip in the same thermal / clock state (N > 1: one such step, a rank's step
    # being ~10 ms).  `op_timing` in the line carries the wall time of these instrumented steps next to the timed ones.
    per_op = None
    n_prof = 0
    ms_with_events = None
    if not args.no_op_timing:
        n_ev = args.steps if world == 1 else 1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_ev):
            _, mss = step(True)
            per_op = mss if per_op is None else [[a + b for a, b in zip(pa, pb)] for pa, pb in zip(per_op, mss)]
            n_prof += 1
        torch.cuda.synchronize()
        ms_with_events = (time.perf_counter() - t1) / n_ev * 1e3
        if use_dist:
            dist.barrier()
    # socket power / clock: sampled over EXTRA untimed steps right after the timed region (rocm-smi queries the SMU; nothing that is not the
    # workload runs next to the timed steps).  Every rank runs the extra steps (they contain the step's collective); rank 0 samples.
    sampler = PowerSampler() if rank == 0 else None
    if sampler is not None:
        sampler.__enter__()
    t_extra = time.perf_counter()
    while True:
        step(False)
        torch.cuda.synchronize()
        more = torch.tensor([1.0 if time.perf_counter() - t_extra < 2.5 else 0.0], device=dev)
        if use_dist:
            dist.all_reduce(more, op=dist.ReduceOp.MIN)        # all ranks leave the loop on the same step
        if float(more.item()) == 0.0:
            break
    if sampler is not None:
        sampler.__exit__()
    per_rank_ms = None
    if use_dist:
        # every rank's own clock over the K timed steps (barrier to barrier), so that the first real SCALE file explains itself: the line's value is the MAX
        tr = torch.zeros(world, device=dev, dtype=torch.float64)
        tr[rank] = elapsed / args.steps * 1e3
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(v), 3) for v in tr.tolist()]
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3

    # the step's collectives on their own (outside the timed region; N > 1 only): the all-gather of the views' maps that every step performs
    # and the drift-guard broadcast of the scene (parallel.sync_scene, ~56 MiB, once per OUTER step -- a second collective beyond
    # north_star's one), so that a scaling projection can carry them
    collectives = None
    if use_dist and shards:
        def timed(fn, n=10):
            fn()
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n * 1e3
        from mvedit_amd import parallel as PL
        force = world == 1
        # the scene as parallel.sync_scene sees it: hash table (2^19 x 12 levels x 2 fp32 = 48 MiB), MLP weights, 128^3 density grid + bitfield
        scene = [torch.zeros(12 << 19, 2, device=dev), torch.zeros(64, 24, device=dev), torch.zeros(4, 64, device=dev),
                 torch.zeros(128 ** 3, device=dev), torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=dev)]
        scene_bytes = sum(t.numel() * t.element_size() for t in scene)
        ag = timed(lambda: PL.all_gather_views(maps_local, V, force=force))
        # ragged shards: V = 9 views after camera pruning (mvedit_3d_pipeline.py:1180-1215) do not divide by the world size -> pad + trim path
        lo9, hi9 = partition_views(9, world, rank)
        maps9 = torch.zeros(hi9 - lo9, 8, 8 * LATENT, 8 * LATENT, device=dev, dtype=torch.float16)
        g9 = PL.all_gather_views(maps9, 9, force=force)
        assert g9.shape[0] == 9
        ag9 = timed(lambda: PL.all_gather_views(maps9, 9, force=force))
        bc = timed(lambda: PL.sync_scene(scene, src=0, force=force))
        tc = torch.tensor([ag, bc, ag9], device=dev, dtype=torch.float64)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        collectives = dict(all_gather_ms=round(float(tc[0]), 3), all_gather_bytes=int(gathered.numel() * gathered.element_size()),
                           all_gather_ragged_9_views_ms=round(float(tc[2]), 3),
                           sync_scene_broadcast_ms=round(float(tc[1]), 3), sync_scene_bytes=int(scene_bytes),
                           note='mvedit_amd.parallel.all_gather_views (32 views; 9 views = ragged shards after camera pruning) and sync_scene; max over '
                                'ranks, 10 back-to-back calls each, outside the timed steps (the 32-view all-gather is also inside them)')

    # ---- roofline of the dominant kernel (rank 0's launches) --------------------------------------------------
    roof = None
    breakdown = {}
    if not args.no_op_timing and per_op is not None:
        for optab, ms_list in zip(optabs, per_op):
            for (ph, cls, fl, lab), m in zip(optab, ms_list):
                d = breakdown.setdefault(cls, dict(ms=0.0, flops=0.0, flops_executed=0.0, launches=0))
                d['ms'] += m / max(n_prof, 1)
                # The plan's flops are ALGORITHMIC: the op as the reference computes it.  Upsample2D runs as four 2 x 2 phase convs
                # (mve_upsample_conv_phases), 4 / 9 of the multiply-adds of the 3 x 3 conv over the upsampled image it is priced at -- like any fast
                # conv algorithm is priced in direct-conv flops; the executed count is kept next to it.
                d['flops'] += fl
                d['flops_executed'] += fl * (4.0 / 9.0 if '(4 phases)' in lab else 1.0)
                d['launches'] += 1
        # The conv and the linear launches run the same kernel (k_gemm_pp, MODE 1 / MODE 0 of one template): it is the dominant
        # kernel by a wide margin, so the roofline object prices ALL of its launches (per-class figures stay in per_class_*).
        gemm = dict(ms=breakdown['conv3x3']['ms'] + breakdown['linear']['ms'], flops=breakdown['conv3x3']['flops'] + breakdown['linear']['flops'],
                    flops_executed=breakdown['conv3x3']['flops_executed'] + breakdown['linear']['flops_executed'],
                    launches=breakdown['conv3x3']['launches'] + breakdown['linear']['launches'])
        dom = 'gemm' if gemm['ms'] >= breakdown['attention']['ms'] else 'attention'
        b = gemm if dom == 'gemm' else breakdown['attention']
        achieved = b['flops'] / (b['ms'] * 1e-3) / 1e12
        traffic = None
        if wl == 'mvedit32':
            if dom == 'gemm':
                tc, tl = pmc_traffic('conv3x3'), pmc_traffic('linear')
                nc, nl = breakdown['conv3x3']['launches'], breakdown['linear']['launches']
                traffic = None if tc is None or tl is None else (tc * nc + tl * nl) / (nc + nl)
            else:
                traffic = pmc_traffic('attention')
        roof = dict(bound='mfma', kernel={'gemm': 'k_gemm_pp: implicit-GEMM conv3x3 (MODE 1) + linear (MODE 0) launches, 256x320 tile, ping-pong loop; the K = 320 '
                                                  'linears are HBM-bound', 'attention': 'k_attention3 / k_attention2 (flash attention)'}[dom],
                    achieved=round(achieved, 1), peak=PEAK_TFLOPS_F16, unit='TFLOP/s', frac=round(achieved / PEAK_TFLOPS_F16, 4),
                    traffic=traffic, traffic_source='static: read from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE '
                    'passes of the default command, 2 x FETCH + WRITE per launch, launch-weighted over the two modes), not measured in this run',
                    launches_per_step=b['launches'], flops_per_step=b['flops'],
                    executed=dict(flops_per_step=b['flops_executed'], tflops_per_s=round(b['flops_executed'] / (b['ms'] * 1e-3) / 1e12, 1),
                                  frac=round(b['flops_executed'] / (b['ms'] * 1e-3) / 1e12 / PEAK_TFLOPS_F16, 4),
                                  note='achieved / frac price every op at the flops of the reference\'s form of it; the three Upsample2D convs execute '
                                       '4/9 of that (four 2x2 phase convs over the source, MVE_UPSAMPLE_PHASES=0 restores the 3x3 form): this is the '
                                       'rate on the multiply-adds actually issued'),
                    avg_launch_ms=round(b['ms'] / b['launches'], 4),
                    note=('since round 6 the durations of the linear launches include the LayerNorm of their output rows (48 per forward: inside the epilogue of the '
                          '320-wide pair tile at the 64x64 level, as the LayerNorm kernel behind the GEMM elsewhere -- mve_gemm_pair_ln); their flops are not counted: '
                          '`frac` is the GEMM / conv flops over a duration that now also normalises (round 5 carried those 2.7 ms in the norm class)') if dom == 'gemm' else None,
                    per_class_ms={k: round(v['ms'], 3) for k, v in breakdown.items()},
                    per_class_tflops={k: round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1) for k, v in breakdown.items() if v['flops'] > 0})

    if rank == 0:
        replicas = world if wl == 'zero123pp' else 1
        total_flops = sum(sum(info['flops'][k] for k in ('conv3x3', 'linear', 'attention')) for info in infos) * world      # (algorithmic, see the roofline block)
        line = {
            'metric': metric,
            'value': round(replicas * 1e3 / ms_per_step, 4), 'unit': 'denoise-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak' if wl == 'zero123pp' else 'strong', 'vs_baseline': None,
            'dtype': 'f16' if dtype == torch.float16 else 'bf16',
            'data': 'synthetic (seeded random ' + ('SD-2.1' if wl == 'zero123pp' else 'SD-1.5') + '-topology weights and latents)',
            'config': {'workload': workload, 'views': 6 if wl == 'zero123pp' else V, 'latent': [120, 80] if wl == 'zero123pp' else LATENT, 'cfg': True,
                       'parallelism': ('replicas only' if wl == 'zero123pp' else f'views/{world}') + (' + all_gather(RGBD/normal maps, 4 MiB per view)' if shards and world > 1 else '')},
            'model_tflops_per_s': round(total_flops / (ms_per_step * 1e-3) / 1e12, 1),
            'model_flops_frac_of_peak': round(total_flops / (ms_per_step * 1e-3) / 1e12 / (PEAK_TFLOPS_F16 * world), 4),
            'rccl_world': dist.get_world_size() if use_dist else 1,
            'roofline': roof,
        }
        if collectives is not None:
            line['collectives'] = collectives
        if per_rank_ms is not None:
            line['per_rank'] = dict(ms_per_step=per_rank_ms, views=[list(partition_views(V, world, r)) for r in range(world)] if shards else None,
                                    note='each rank\'s own wall clock per timed step (barrier to barrier; the step ends with the all-gather, so ranks agree to '
                                         'the skew of its completion); `ms_per_step` of the line is the max')
        if sampler is not None and sampler.summary() is not None:
            line['power'] = sampler.summary()
            # energy per unit of algorithmic work over the sampled (untimed) steps: what a kernel change has to lower when the package sits at its
            # power cap (rocm-smi --showmaxpower on this pool: 1400 W, profiles/r04_smi_maxpower.log)
            line['power']['joule_per_step'] = round(line['power']['avg_w'] * ms_per_step * 1e-3, 2)
            line['power']['joule_per_tflop'] = round(line['power']['avg_w'] * ms_per_step * 1e-3 / (total_flops / 1e12), 3)
            line['power']['cap_w'] = 1400
        if world == 1 and roof is not None:
            try:
                clk = clock_probe(dev)
                clk['frac_at_clock'] = round(roof['achieved'] / clk['peak_at_clock'], 4)
                # which limiter holds the clock there: the SMU's package-power tracking -- gpu_metrics ppt_residency_acc advances only inside the
                # conv loop, thermal / prochot residencies stay 0 (tools/limiter_probe.py, profiles/r05_limiter.log).  Diagnosis, not a score.
                clk['limiter'] = 'PPT (package power tracking): profiles/r05_limiter.log'
                roof['clock'] = clk
            except Exception as e:      # informational only
                roof['clock'] = {'error': repr(e)[:200]}
        if ms_with_events is not None:
            line['op_timing'] = dict(ms_per_step_timed_region=round(ms_per_step, 3), ms_per_step_with_per_op_events=round(ms_with_events, 3), steps_with_events=n_prof,
                                     note='the timed K steps carry no events; the roofline / per-class durations are HIP events around every launch over the same '
                                          'steps repeated right after the timed region (the events cost ~2 ms of host + queue time per 64-image step)')
        if world == 1 and not args.no_cpu_baseline:
            hw = (120, 80) if wl == 'zero123pp' else (LATENT, LATENT)
            if timed_rows is not None:
                # the checker reads ROWS OF THE TIMED BATCH: the fp32 oracle runs the two items alone (B = 2, ~2 x 3 s per
                # repeat on the host cores -- also the cpu_baseline sample), the engine's rows come out of the last timed 64-image step, i.e. from
                # the launch decisions only that batch takes (un-split accumulation chains, reduced slice counts: csrc/gemm.hip launch_gemm)
                ridx, got_rows = timed_rows
                x_all, _, c_all, _, _ = passes[0]
                base, (bx, bctx, bout) = cpu_baseline(cfg, dtype, hw, x=x_all[ridx], ctx=c_all[ridx], repeats=2)
            else:
                base, (bx, bctx, bout) = cpu_baseline(cfg, dtype, hw)
            nfw = 2 if wl == 'zero123pp' else forwards      # (zero123pp: the 40x40 condition pass is ~1/6 of the main pass; counted as part of it)
            base['value'] = 1.0 / (nfw * base['seconds_per_forward'])
            base['unit'] = 'denoise-steps/s' + ('' if wl == 'zero123pp' else ' (32 views)')
            base['sample'] += f' ({nfw})'
            line['cpu_baseline'] = base
            # the oracle as checker (north_star: 1e-3 rel fp16)
            try:
                if timed_rows is not None:
                    per_row = [float((got_rows[k] - bout[k]).norm() / bout[k].norm()) for k in range(len(ridx))]
                    rel = float((got_rows - bout).norm() / bout.norm())
                    line['parity'] = dict(rel_l2_vs_fp32_oracle=round(rel, 6), per_row=[round(v, 6) for v in per_row], rows_of_the_timed_batch=ridx,
                                          shape=[len(ridx), 4, hw[0], hw[1]], batch=int(x_all.shape[0]), north_star_bar=1e-3,
                                          within_bar=bool(max(per_row) <= 1e-3),
                                          note='rows of the last TIMED step\'s UNet output (view 0: unconditional and text row of the 2V-image batch) against '
                                               'fp32 oracle forwards of the same two items over the same 16-bit weights; tests/test_unet.py holds the same '
                                               'comparison at B = 64 and the per-kernel 1e-3 bars')
                else:
                    eng._set_attention(None, 1, hw[0], hw[1])
                    got = eng._run(0, bx.to(dev, dtype), torch.full((1,), 499.0, device=dev), bctx.to(dev, dtype), 1, None, None, None).float().cpu()
                    rel = float((got - bout).norm() / bout.norm())
                    line['parity'] = dict(rel_l2_vs_fp32_oracle=round(rel, 6), shape=[1, 4, hw[0], hw[1]], north_star_bar=1e-3,
                                          note='one forward at the benchmark latent size; the oracle runs fp32 arithmetic over the same 16-bit weights')
            except Exception as e:
                line['parity'] = {'error': repr(e)[:300]}
        line['config']['residual_stream'] = ('(hi, lo8) pair: the 16-bit matrix-core operand + an 8-bit E5M2 remainder (the engine\'s default mode; end-to-end error inside north_star\'s 1e-3)'
                                             if pair_mode else '16-bit, rounded after every block (the reference\'s half modules)')
        phases = os.environ.get('MVE_UPSAMPLE_PHASES', '1') != '0'
        line['config']['upsample2d'] = ('four 2x2 phase convs over the source (summed 3x3 taps rounded once to the storage type; 4/9 of the multiply-adds)'
                                        if phases else '3x3 conv over the nearest-upsampled image (as the reference)')
        if world == 1 and wl == 'mvedit32' and not args.no_extra:
            # what one rank of an N-GPU job runs (32 views x CFG / N images), on this box: the projection the first real SCALE file is to be read against
            try:
                line['scaling_projection'] = scaling_projection(eng, dev, dtype, ms_per_step)
            except Exception as e:
                line['scaling_projection'] = {'error': repr(e)[:300]}
        if world == 1 and not args.no_extra and wl == 'mvedit32' and dtype == torch.float16 and pair_mode:
            # the other configurations BASELINE.json names, and the two other numeric modes, as compact driver-visible lines (3 steps each)
            del eng
            torch.cuda.empty_cache()
            extras = []
            head_ref = (bx, bctx, bout, 1) if (not args.no_cpu_baseline and 'parity' in line and 'error' not in line['parity']) else None
            for (wl2, dt2, pair2) in (('mvedit32', torch.float16, False), ('use_reference', torch.float16, True), ('zero123pp', torch.float16, True),
                                      ('mvedit32', torch.bfloat16, True)):
                try:
                    ref = None
                    if not args.no_cpu_baseline:
                        if wl2 == 'mvedit32' and dt2 == torch.float16:
                            ref = head_ref           # (the two rows of the headline's checker, here as a B = 2 forward of the plain-stream engine)
                        elif wl2 == 'mvedit32':
                            ref = cpu_baseline(dict(SD15_CONFIG), dt2, (LATENT, LATENT), repeats=1)[1] + (1,)
                        elif wl2 == 'use_reference':
                            ref = cpu_baseline(dict(SD15_CONFIG), dt2, (LATENT, LATENT), B=2, n_img=2, repeats=1)[1] + (2,)
                        elif wl2 == 'zero123pp':
                            zp = make_passes('zero123pp', dict(SD21_CONFIG), 6, 0, 6, 6, 1, dev, dt2)[0]
                            ref = cpu_baseline(dict(SD21_CONFIG), dt2, zero123pp_passes=zp)[1] + (1,)
                    extras.append(measure_workload(dev, wl2, dt2, pair2, parity_ref=ref))
                except Exception as e:
                    extras.append({'workload': wl2, 'error': repr(e)[:300]})
            line['extra_workloads'] = extras
            try:
                line['outer_step'] = outer_step(dev)
            except Exception as e:
                line['outer_step'] = {'error': repr(e)[:300]}
        if world == 1 and not args.no_secondary and wl == 'mvedit32':
            try:
                line['secondary'] = secondary(dev)
            except Exception as e:      # a secondary figure must never take the headline line with it
                line['secondary'] = {'error': repr(e)[:300]}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
