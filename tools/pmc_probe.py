"""Tiny workload for PMC passes: a few launches of the benchmark's heavy GEMM / conv / attention shapes (64 images)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops  # noqa: E402

dt, dev = torch.float16, 'cuda'
B = 64
for (H, C1, Cout) in [(64, 320, 320), (16, 1280, 1280)]:
    x = torch.randn(B * H * H, C1, device=dev, dtype=dt)
    wc = torch.randn(Cout, C1 // 64, 3, 3, 64, device=dev, dtype=dt)
    for _ in range(3):
        ops.conv3x3(x, wc, B, H, H, flags=ops.W_CHUNK64, splitk=True)
for (M, N, K, fl) in [(B * 4096, 2560, 320, ops.GEGLU), (B * 4096, 320, 320, 0), (B * 1024, 640, 2560, 0)]:
    a = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt)
    for _ in range(3):
        ops.gemm(a, w, flags=fl)
qkv = torch.randn(16 * 4096, 960, device=dev, dtype=dt)
for _ in range(3):
    ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], 16, 4096, 4096, 8, 40)
torch.cuda.synchronize()
