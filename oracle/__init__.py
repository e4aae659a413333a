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


DEVCORE = os.path.join(HERE, 'libdevcore.so')


def build_devcore(force=False):
    """Host build (g++) of the product's host/device core headers behind plain loops (devcore_host.cpp): lets the CPU tests run the
    per-pixel arithmetic of kernels whose first GPU run is pending."""
    src = os.path.join(HERE, 'devcore_host.cpp')
    core = os.path.join(os.path.dirname(HERE), 'mvedit_amd', 'csrc')
    deps = [src] + [os.path.join(core, h) for h in ('raster_grad_core.h', 'shading_core.h', 'recon_loss_core.h', 'mesh_reg_core.h', 'mesh_loss_core.h', 'blur_core.h', 'sh_core.h')]
    if force or not os.path.exists(DEVCORE) or any(os.path.getmtime(d) > os.path.getmtime(DEVCORE) for d in deps):
        subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-Wall', '-shared', '-o', DEVCORE, src, '-lm'],
                       check=True, capture_output=True)
    return DEVCORE
