"""Workload of tools/pmc_attention.sh: three launches of the level-0 self-attention (16 images) under one mve_attention_tune variant."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvedit_amd import _lib, ops  # noqa: E402

_lib.raw('mve_attention_tune')(int(sys.argv[1]))
qkv = torch.randn(16 * 4096, 960, device='cuda', dtype=torch.float16)
for _ in range(3):
    ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], 16, 4096, 4096, 8, 40)
torch.cuda.synchronize()
