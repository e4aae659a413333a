"""One Zero123++ denoise step (BASELINE config 2: condition pass + tiled-view pass, CFG pair) a few times back to back: workload of a kernel trace."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import synthetic as U  # noqa: E402
from mvedit_amd.unet import SD21_CONFIG, UNet2DConditionEngine  # noqa: E402
from tools.bench_parts import CTX_LEN, make_passes  # noqa: E402

dev, dtype = torch.device('cuda', 0), torch.float16
cfg = dict(SD21_CONFIG)
eng = UNet2DConditionEngine.from_state_dict(U.make_state_dict(cfg, seed=1234, dtype=dtype), cfg, dtype, dev)
passes = make_passes('zero123pp', cfg, 6, 0, 6, 6, 1, dev, dtype)[0]


def step():
    for (x_, t_, c_, n_, kw_) in passes:
        eng._set_attention(kw_, x_.shape[0], x_.shape[2], x_.shape[3])
        eng._run(0, x_, t_, c_, n_, None, None, None)


for _ in range(3):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
ev[0].record()
for i in range(4):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
print('step ms:', ' '.join(f'{ev[i].elapsed_time(ev[i + 1]):.3f}' for i in range(4)))
