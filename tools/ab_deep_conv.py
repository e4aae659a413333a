"""The deep-level (8x8, 16x16 latents) convolutions of the 64-image forward under the dispatcher's alternatives."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, _lib
from tools.microbench import timeit
tune = _lib.raw('mve_gemm_tune')
dt, dev = torch.float16, 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for (H, C1, Cout) in [(8, 1280, 1280), (8, 2560, 1280), (16, 1280, 1280), (16, 2560, 1280)]:
    x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
    w = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt) * (9 * C1) ** -0.5
    fl = 2 * B * H * H * Cout * 9 * C1
    ref = None
    for name, t, sk in (('production (tune 256), split-K on', 256, True), ('production, split-K off', 256, False), ('big tile always (tune 1), split-K on', 1, True),
                        ('big tile always, split-K off', 1, False), ('small kernel (tune 0), split-K on', 0, True), ('small kernel, split-K off', 0, False),
                        ('production, real split-K + reducer (no SEQ)', 256 | (1 << 29), True)):
        tune(t)
        f = lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=sk)[0]
        o = f()
        if ref is None: ref = o
        ms = min(timeit(f, 2, 6) for _ in range(2)) * 1e3
        print(f'conv B={B} H={H:2d} {C1}->{Cout}: {name:48s} {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF  equal_to_first={torch.equal(o, ref)}', flush=True)
tune(256)
