"""The forward one rank of an N-GPU job runs (B images, default 8), a few times back to back: the workload of tools/trace_gaps.sh."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import synthetic as U  # noqa: E402
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = UNet2DConditionEngine(SD15_CONFIG, torch.float16)
g = torch.Generator().manual_seed(0)
eng.load_state_dict({n: torch.randn(sh, generator=g, dtype=torch.float16) * 0.02 for n, sh in U.param_shapes(SD15_CONFIG).items()})
x = torch.randn(B, 4, 64, 64, device='cuda', dtype=torch.float16)
ctx = torch.randn(B, 77, 768, device='cuda', dtype=torch.float16)
for _ in range(3):
    eng(x, 499, ctx)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    eng(x, 499, ctx)
    ev[i + 1].record()
torch.cuda.synchronize()
print('forward ms:', ' '.join(f'{ev[i].elapsed_time(ev[i + 1]):.3f}' for i in range(reps)))
