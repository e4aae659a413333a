"""Section timing of the two-blocks-per-CU tile (k_gemm_pp<..., 160, 3 slots>) next to the 320-wide tile on the same shapes."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, _lib
from tools.microbench import timeit
prof = _lib.raw('mve_gemm_pp_profile'); prof.argtypes = [ctypes.c_void_p]
tune = _lib.raw('mve_gemm_tune')
dt, dev = torch.float16, 'cuda'
B = 64
def report(name, f, nblocks, nsteps, flops, per_cu):
    prof(None)
    ms = timeit(f, 2, 5) * 1e3
    buf = torch.zeros(nblocks * 64, dtype=torch.int64, device=dev)
    prof(ctypes.c_void_p(buf.data_ptr()))
    f(); torch.cuda.synchronize()
    ms_p = timeit(f, 1, 3) * 1e3
    prof(None)
    vv = buf.view(nblocks, 8, 8).double().cpu()
    v = vv / nsteps
    g0, g1 = v[:, :4].mean((0, 1)), v[:, 4:].mean((0, 1))
    pro, epi = float(vv[:, :, 6].mean()), float(vv[:, :, 7].mean())
    loop = float(vv[:, :, :4].sum(-1).mean())
    print(f'{name}: {ms:.3f} ms = {flops / ms / 1e9:.0f} TF (instrumented {ms_p:.3f}); per tile: prologue {pro:.0f} K loop {loop:.0f} epilogue {epi:.0f} cycles ({nsteps} steps, {per_cu} blocks/CU); per step '
          f'g0: L {g0[0]:.0f} (reads {g0[4]:.0f} prep {g0[5]:.0f}) waitL {g0[1]:.0f} M {g0[2]:.0f} waitM {g0[3]:.0f} sum {float(g0[:4].sum()):.0f} | '
          f'g1: L {g1[0]:.0f} (reads {g1[4]:.0f} prep {g1[5]:.0f}) waitL {g1[1]:.0f} M {g1[2]:.0f} waitM {g1[3]:.0f} sum {float(g1[:4].sum()):.0f}', flush=True)
for (M, N, K, fl, res) in [(B * 4096, 320, 1280, 0, 1), (B * 4096, 2560, 320, 1, 0), (B * 1024, 1920, 640, 0, 0), (B * 256, 3840, 1280, 0, 0)]:
    a = torch.randn(M, K, device=dev, dtype=dt); w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    bias = torch.randn(N, device=dev, dtype=torch.float32)
    r = torch.randn(M, N, device=dev, dtype=dt) if res else None
    f = lambda: ops.gemm(a, w, bias=bias, residual=r, flags=ops.GEGLU if fl else 0, rows_per_image=0)
    tune(1)
    report(f'320-wide M={M} N={N} K={K} geglu={fl} res={res}', f, (M // 256) * (N // 320), K // 32, 2 * M * N * K, 1)
    tune(1 | (1 << 26))
    report(f'2 x 160  M={M} N={N} K={K} geglu={fl} res={res}', f, (M // 256) * (N // 160), K // 32, 2 * M * N * K, 2)
tune(256)
