"""Per-rank forward time at the batch a rank sees with N GPUs (32 views x CFG / N), same box: the strong-scaling preview."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import synthetic as U
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine
from tools.microbench import timeit
eng = UNet2DConditionEngine(SD15_CONFIG, torch.float16)
g = torch.Generator().manual_seed(0)
eng.load_state_dict({n: torch.randn(sh, generator=g, dtype=torch.float16) * 0.02 for n, sh in U.param_shapes(SD15_CONFIG).items()})
base = None
for n_gpu in (1, 2, 4, 8):
    B = 64 // n_gpu
    x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
    ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
    eng(x, 499, ctx)
    t = min(timeit(lambda: eng(x, 499, ctx), 1, 4) for _ in range(2)) * 1e3
    base = base or t
    print(f'N={n_gpu} images/rank={B:2d}: {t:7.2f} ms  -> speedup {base / t:4.2f}x of {n_gpu}', flush=True)

# same sweep with the opt-in hipGraph replay (mve_unet_graph): the forward is captured on its second call with identical tensors
if '--graph' in sys.argv:
    eng.enable_graph(True)
    side = torch.cuda.Stream()                     # stream capture is not allowed on the legacy default stream
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
    for n_gpu in (1, 2, 4, 8):
        B = 64 // n_gpu
        x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
        ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
        tt = torch.full((B,), 499.0, device='cuda')
        ref = eng(x, tt, ctx)[0].clone()
        eng(x, tt, ctx)
        eng(x, tt, ctx)                                   # captured by now if the output buffer address repeated
        t = min(timeit(lambda: eng(x, tt, ctx), 1, 4) for _ in range(2)) * 1e3
        same = torch.equal(eng(x, tt, ctx)[0], ref)
        print(f'graph N={n_gpu} images/rank={B:2d}: {t:7.2f} ms   replay == eager: {same}', flush=True)
