"""ORACLE -- test infrastructure only.

Builds the REFERENCE's own ray-marching extension for the MI355X from the
sources where they lie under /root/reference (never copied into this repo):

    /root/reference/lib/ops/raymarching/src/{raymarching.cu,bindings.cpp}

exactly as the reference's lib/ops/raymarching/backend.py:32-39 does
(`torch.utils.cpp_extension.load`), which on a ROCm PyTorch hipifies the .cu on
the fly.  hipify writes its translated copy next to the source it reads, and
/root/reference is read-only, so the two files are staged in a throw-away
directory under /tmp; only the resulting python extension module lands in
oracle/_ref/ (git-ignored, travels to the GPU box with the snapshot).

The module is used by tests/test_raymarching_ref.py to pin
oracle/raymarching_oracle.c (and transitively the HIP kernels) against the
reference's own kernels running on the same GPU.  It is compiled with
-ffp-contract=off so that it evaluates the source expressions as written --
the same convention as the oracle and the product kernels -- which makes
bit-exact comparison of index buffers meaningful.

Run:  python oracle/build_ref.py        (needs /root/reference; no GPU needed)
"""
import glob
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
REF_SRC = '/root/reference/lib/ops/raymarching/src'
NAME = '_raymarching_ref'
# the reference's second native op, lib/ops/shencoder (pins csrc/sh.hip the same way: tests/test_shencoder.py)
SH_SRC = '/root/reference/lib/ops/shencoder/src'
SH_NAME = '_shencoder_ref'


def built_module_path(name=NAME):
    hits = glob.glob(os.path.join(OUT, name + '*.so'))
    return hits[0] if hits else None


def build_shencoder(verbose=False):
    """same recipe for lib/ops/shencoder/src/{shencoder.cu,bindings.cpp} (lib/ops/shencoder/backend.py)"""
    if not os.path.isdir(SH_SRC) or built_module_path(SH_NAME):
        return built_module_path(SH_NAME)
    os.environ.setdefault('PYTORCH_ROCM_ARCH', 'gfx950')
    from torch.utils.cpp_extension import load
    os.makedirs(OUT, exist_ok=True)
    stage = tempfile.mkdtemp(prefix='mve_ref_stage_')
    build_dir = tempfile.mkdtemp(prefix='mve_ref_build_')
    try:
        for f in ('shencoder.cu', 'shencoder.h', 'bindings.cpp'):
            shutil.copy(os.path.join(SH_SRC, f), os.path.join(stage, f))
        load(name=SH_NAME, sources=[os.path.join(stage, 'shencoder.cu'), os.path.join(stage, 'bindings.cpp')],
             extra_cflags=['-O3', '-std=c++17'], extra_cuda_cflags=['-O3', '-std=c++17', '-ffp-contract=off'],
             build_directory=build_dir, verbose=verbose, is_python_module=False)
        so = glob.glob(os.path.join(build_dir, SH_NAME + '*.so'))
        assert so, 'extension build produced no .so'
        shutil.copy(so[0], os.path.join(OUT, os.path.basename(so[0])))
    finally:
        shutil.rmtree(stage, ignore_errors=True)
        shutil.rmtree(build_dir, ignore_errors=True)
    return built_module_path(SH_NAME)


def build(verbose=False):
    if not os.path.isdir(REF_SRC):
        return built_module_path()  # GPU box: use the prebuilt file if it travelled
    if built_module_path():
        return built_module_path()
    os.environ.setdefault('PYTORCH_ROCM_ARCH', 'gfx950')
    from torch.utils.cpp_extension import load
    os.makedirs(OUT, exist_ok=True)
    stage = tempfile.mkdtemp(prefix='mve_ref_stage_')
    build_dir = tempfile.mkdtemp(prefix='mve_ref_build_')
    try:
        for f in ('raymarching.cu', 'raymarching.h', 'bindings.cpp'):
            shutil.copy(os.path.join(REF_SRC, f), os.path.join(stage, f))
        load(name=NAME,
             sources=[os.path.join(stage, 'raymarching.cu'), os.path.join(stage, 'bindings.cpp')],
             extra_cflags=['-O3', '-std=c++17'],
             extra_cuda_cflags=['-O3', '-std=c++17', '-ffp-contract=off'],
             build_directory=build_dir, verbose=verbose, is_python_module=False)
        so = glob.glob(os.path.join(build_dir, NAME + '*.so'))
        assert so, 'extension build produced no .so'
        shutil.copy(so[0], os.path.join(OUT, os.path.basename(so[0])))
    finally:
        shutil.rmtree(stage, ignore_errors=True)
        shutil.rmtree(build_dir, ignore_errors=True)
    return built_module_path()


def load_module(name=NAME):
    """Import the prebuilt reference extension (requires a GPU process with torch imported)."""
    import importlib.util
    import torch  # noqa: F401
    path = built_module_path(name)
    if path is None:
        return None
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
    print(build_shencoder(verbose='-v' in sys.argv))
