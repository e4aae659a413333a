"""Tiny workload for PMC passes: a few launches of the big GEMM / conv / attention shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import ops  # noqa: E402

dt, dev = torch.float16, 'cuda'
a = torch.randn(8192, 8192, device=dev, dtype=dt)
w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(3):
    ops.gemm(a, w)
x = torch.randn(32 * 64 * 64, 320, device=dev, dtype=dt)
wc = torch.randn(320, 5, 3, 3, 64, device=dev, dtype=dt)
for _ in range(3):
    ops.conv3x3(x, wc, 32, 64, 64, flags=ops.W_CHUNK64)
qkv = torch.randn(16 * 4096, 960, device=dev, dtype=dt)
for _ in range(3):
    ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], 16, 4096, 4096, 8, 40)
torch.cuda.synchronize()
