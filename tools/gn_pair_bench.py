"""GroupNorm(+SiLU) on stream pairs vs plain tensors at the UNet's level-0 / level-1 shapes (64 images): ms per call and the effective TB/s on the bytes
each form has to move (plain: 2 reads + 1 write of 2 B; pair: 2 reads of 3 B + 1 write of 2 B)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib, ops  # noqa: E402
from mvedit_amd.ops import dt as _dt  # noqa: E402

dev = torch.device('cuda:0')
G = 32
SHAPES = [(64, 4096, 320, 0), (64, 4096, 640, 320), (64, 4096, 640, 0), (64, 1024, 640, 0), (64, 1024, 1280, 0), (64, 1024, 640, 320), (64, 1024, 1280, 640), (64, 256, 1280, 0), (64, 256, 1280, 640),
          (64, 256, 1280, 1280), (64, 64, 1280, 1280)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(a) for a in sys.argv[1:5])]
for (B, HW, C1, C2) in SHAPES:
    C = C1 + C2
    g = torch.Generator().manual_seed(1)
    mk = lambda c: ops.split_pair(torch.randn(B * HW, c, generator=g), torch.float16)
    (h1, l1), (h2, l2) = mk(C1), (mk(C2) if C2 else (None, None))
    h1, l1 = h1.to(dev), l1.to(dev)
    h2, l2 = (h2.to(dev), l2.to(dev)) if C2 else (None, None)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    ws = torch.empty(_lib.raw('mve_groupnorm_workspace_bytes')(B, HW, C, G), dtype=torch.uint8, device=dev)
    out = torch.empty(B * HW, C, dtype=torch.float16, device=dev)

    def run(pair):
        _lib.call('mve_groupnorm_silu_pair', _dt(h1), _lib.ptr(h1), C1, _lib.ptr(h2), C2, B, HW, G, 1e-5, _lib.ptr(gam), _lib.ptr(bet), 1,
                  _lib.ptr(out), _lib.ptr(ws), _lib.ptr(l1) if pair else None, (_lib.ptr(l2) if C2 else None) if pair else None, _lib.stream_ptr(dev))
    res = []
    for pair in (False, True):
        for _ in range(3):
            run(pair)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run(pair)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        nbytes = B * HW * C * ((3 + 3 + 2) if pair else (2 + 2 + 2))
        res.append((ms, nbytes / ms / 1e9))
    print(f'B={B} HW={HW:4d} C={C1}+{C2}: plain {res[0][0]:.3f} ms ({res[0][1]:.2f} TB/s)   pair {res[1][0]:.3f} ms ({res[1][1]:.2f} TB/s)', flush=True)
