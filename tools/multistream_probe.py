"""Small-batch forwards as S concurrent sub-batches on S HIP streams (one plan, S workspaces): does the chip fill when a rank's 8 images run as
2 x 4 or 4 x 2 side by side?  Same box, same weights; results must be bit-identical to the single-stream forward (batch invariance among small
batches).  Usage: python tools/multistream_probe.py [B ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import synthetic as U  # noqa: E402
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine  # noqa: E402

eng = UNet2DConditionEngine(SD15_CONFIG, torch.float16)
g = torch.Generator().manual_seed(0)
eng.load_state_dict({n: torch.randn(sh, generator=g, dtype=torch.float16) * 0.02 for n, sh in U.param_shapes(SD15_CONFIG).items()})
if '--pair' in sys.argv:
    eng.set_residual_pair(True)
Bs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 16, 2]


def wall(fn, warm=2, it=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(it):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / it)
    return best * 1e3


for B in Bs:
    x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
    ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
    ref = eng(x, 499, ctx)[0].clone()
    base = wall(lambda: eng(x, 499, ctx))
    print(f'B={B:2d} one stream: {base:7.3f} ms', flush=True)
    for S in (2, 4, 8):
        if B % S or B // S < 1:
            continue
        b = B // S
        streams = [torch.cuda.Stream() for _ in range(S)]
        info = eng.plan(b, 64, 64, 77)
        wss = [torch.empty(info['workspace_bytes'], dtype=torch.uint8, device='cuda') for _ in range(S)]
        out = torch.empty_like(ref)
        xs, cs = x.split(b), ctx.split(b)
        cur = torch.cuda.current_stream()

        def run():
            ev = torch.cuda.Event()
            ev.record(cur)
            for i, s in enumerate(streams):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    eng._set_attention(None, b, 64, 64)
                    eng._run(0, xs[i], 499, cs[i], 1, None, None, out[i * b:(i + 1) * b], workspace=wss[i])
            for s in streams:
                cur.wait_stream(s)
        run()
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        t = wall(run)
        print(f'B={B:2d} as {S} x {b} on {S} streams: {t:7.3f} ms  ({base / t:4.2f}x)  bit-identical to one stream: {same}', flush=True)
