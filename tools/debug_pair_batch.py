"""Does a row of a large batch equal the same row computed in a small batch?  (round 5 bisect: the pair mode's 64-image rows were ~100 % off the
oracle while B = 2 forwards were inside 1e-3.)  Prints, per batch size, the rel-L2 between rows {0, B/2, B-1} of the batch and the same three
items run alone, under the dispatcher switches given in MVE_DEBUG_TUNE (mve_gemm_tune word) -- environment switches (MVE_UPSAMPLE_PHASES,
MVE_RESIDUAL_PAIR, MVE_GEMM_STRICT_SPLITK ...) are read by the library at load, so the caller loops over processes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib, synthetic as U  # noqa: E402
from mvedit_amd.unet import SD15_CONFIG, UNet2DConditionEngine  # noqa: E402

word = os.environ.get('MVE_DEBUG_TUNE')
if word is not None:
    _lib.raw('mve_gemm_tune')(int(word, 0))
eng = UNet2DConditionEngine.from_state_dict(U.make_state_dict(dict(SD15_CONFIG), seed=1234, dtype=torch.float16), dict(SD15_CONFIG), torch.float16, 'cuda')
tag = ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('MVE_'))
g = torch.Generator().manual_seed(5)
x = torch.randn(64, 4, 64, 64, generator=g).half().cuda()
ctx = torch.randn(64, 77, 768, generator=g).half().cuda()
for B in [int(a) for a in sys.argv[1:]] or [64, 32, 16]:
    rows = [0, B // 2, B - 1]
    full = eng(x[:B], 499, ctx[:B])[0].float()
    alone = eng(x[:B][rows].contiguous(), 499, ctx[:B][rows].contiguous())[0].float()
    d = [float((full[r] - alone[k]).norm() / alone[k].norm()) for k, r in enumerate(rows)]
    print(f'[{tag}] pair={eng.residual_pair} B={B:2d}: rows {rows} vs alone: ' + ' '.join(f'{v:.3e}' for v in d) + f'   finite={bool(torch.isfinite(full).all())}', flush=True)
