"""Section timing of the ping-pong GEMM / conv main loop (csrc/gemm_pp.hip, mve_gemm_pp_profile): average shader clocks per K step of
the four parts of a step, per wave group, next to the wall time of the uninstrumented kernel."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops, _lib  # noqa: E402
from tools.microbench import timeit  # noqa: E402

prof = _lib.raw('mve_gemm_pp_profile')
prof.argtypes = [ctypes.c_void_p]
tune = _lib.raw('mve_gemm_tune')
tune(1)
dt, dev = torch.float16, 'cuda'
B = 64


def report(name, f, nblocks, nsteps, flops):
    prof(None)
    ms = timeit(f, 2, 5) * 1e3
    buf = torch.zeros(nblocks * 64, dtype=torch.int64, device=dev)
    prof(ctypes.c_void_p(buf.data_ptr()))
    f()
    torch.cuda.synchronize()
    ms_p = timeit(f, 1, 3) * 1e3
    prof(None)
    v = buf.view(nblocks, 8, 8).double().cpu() / nsteps
    g0, g1 = v[:, :4].mean((0, 1)), v[:, 4:].mean((0, 1))
    tot0, tot1 = float(g0[:4].sum()), float(g1[:4].sum())
    vv = buf.view(nblocks, 8, 8).double().cpu()
    pro, epi = float(vv[:, :, 6].mean()), float(vv[:, :, 7].mean())
    loop = float(vv[:, :, :4].sum(-1).mean())
    ncu = 256
    clk = (pro + loop + epi) * ((nblocks + ncu - 1) // ncu) / (ms_p * 1e-3) / 1e9
    print(f'   per tile: prologue {pro:.0f}  K loop {loop:.0f}  epilogue {epi:.0f} cycles;  tiles per CU {(nblocks + ncu - 1) // ncu};  => shader clock >= {clk:.2f} GHz (instrumented wall)')
    print(f'{name}: {ms:.3f} ms = {flops / ms / 1e9:.0f} TF (instrumented {ms_p:.3f} ms), {nsteps} steps; cycles per step  '
          f'group0 L {g0[0]:.0f} (reads {g0[4]:.0f} prep {g0[5]:.0f}) waitL {g0[1]:.0f} M {g0[2]:.0f} waitM {g0[3]:.0f} sum {tot0:.0f} | '
          f'group1 L {g1[0]:.0f} (reads {g1[4]:.0f} prep {g1[5]:.0f}) waitL {g1[1]:.0f} M {g1[2]:.0f} waitM {g1[3]:.0f} sum {tot1:.0f} | MFMA floor 640 per section', flush=True)


for (H, C1, Cout) in ([] if (len(sys.argv) > 1 and sys.argv[1] == 'deep') else [(64, 320, 320), (32, 640, 640), (16, 1280, 1280)]):
    x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
    w = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt) * (9 * C1) ** -0.5
    f = lambda: ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=False)
    report(f'conv H={H} {C1}->{Cout}', f, (B * H * H // 256) * (Cout // 320), 9 * C1 // 32, 2 * B * H * H * Cout * 9 * C1)
shapes = [(B * 4096, 2560, 320, 1), (B * 4096, 320, 1280, 0), (B * 1024, 640, 2560, 0), (B * 256, 3840, 1280, 0), (B * 256, 1280, 1280, 0), (B * 1024, 640, 640, 0)]
if len(sys.argv) > 1 and sys.argv[1] == 'deep':
    shapes = shapes[-2:]
for (M, N, K, fl) in shapes:
    a = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt) * K ** -0.5
    f = lambda: ops.gemm(a, w, flags=ops.GEGLU if fl else 0, rows_per_image=0)
    report(f'gemm M={M} N={N} K={K}', f, (M // 256) * (N // 320), K // 32, 2 * M * N * K)
    if not fl:      # the same launch with the epilogue's loads: bias + residual (attn.to_out / ff.out of the UNet)
        bias = torch.randn(N, device=dev, dtype=torch.float32)
        res = torch.randn(M, N, device=dev, dtype=dt)
        f = lambda: ops.gemm(a, w, bias=bias, residual=res, rows_per_image=0)
        report(f'gemm M={M} N={N} K={K} +bias+residual', f, (M // 256) * (N // 320), K // 32, 2 * M * N * K)
