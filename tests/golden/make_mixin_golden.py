"""Writes tests/golden/mixin_ref.npz by EXECUTING the reference's `Adapter3DMixin.get_noise_pred`
(lib/pipelines/adapter3d_mixin.py:68-135, the method's source taken from the file with `ast`; the module itself cannot be imported:
diffusers / mmcv are absent) over the deterministic stand-in UNet / ControlNet of tests/stubs.py.  Nothing is copied into the repo.
Run from the repo root (needs /root/reference):  python tests/golden/make_mixin_golden.py"""
import ast
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import stubs  # noqa: E402

REF = '/root/reference/lib/pipelines/adapter3d_mixin.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mixin_ref.npz')


def reference_method():
    tree = ast.parse(open(REF).read())
    ns = dict(torch=torch)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == 'Adapter3DMixin':
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == 'get_noise_pred':
                    exec(compile(ast.Module([fn], []), REF, 'exec'), ns)
    return ns['get_noise_pred']


def main():
    fn = reference_method()

    class Pipe:
        unet, controlnet = stubs.StubUNet(), stubs.StubControlNet()
    out = {}
    with torch.no_grad():
        for name, kw in stubs.cases().items():
            out[name] = fn(Pipe(), **kw).numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
