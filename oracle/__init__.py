"""ORACLE -- test infrastructure only.

CPU restatements of the reference algorithms used to check the HIP path:
  raymarching_oracle.c / raymarching.py   plain C, follows lib/ops/raymarching/src/raymarching.cu
  unet_oracle.py                          plain torch fp32, follows diffusers==0.27.2 UNet2DConditionModel
                                          as wrapped by lib/models/architecture/diffusers.py
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'liboracle.so')


def build(force=False):
    """Compile the C restatement with gcc (Makefile in this directory)."""
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('_oracle.c')]
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
    if stale:
        subprocess.run(['make', '-C', HERE, '-B', 'liboracle.so'], check=True, capture_output=True)
    return LIB
