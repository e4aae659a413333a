import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, _lib
from tools.microbench import timeit
dt, dev = torch.float16, 'cuda'
B = 64
for (M, N, K, rpi) in [(B * 4096, 2560, 320, 4096), (B * 1024, 5120, 640, 1024), (B * 256, 10240, 1280, 256)]:
    a = torch.randn(M, K, device=dev, dtype=dt); w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    bias = torch.randn(N, device=dev, dtype=torch.float32)
    for rnd in range(2):
        for name, fl in (('geglu', ops.GEGLU), ('geglu-no-gelu', ops.GEGLU | (1 << 20)), ('plain N (2x output)', 0)):
            f = lambda: ops.gemm(a, w, bias=bias, flags=fl, rows_per_image=rpi)
            f(); t = timeit(f, 2, 8) * 1e3
            print(f'M={M} N={N} K={K} {name:22s} {t:.3f} ms  {2*M*N*K/t/1e9:6.0f} TF', flush=True)
