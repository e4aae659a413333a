"""Writes tests/golden/mixin_ref.npz by EXECUTING the reference's `Adapter3DMixin.get_noise_pred`
(lib/pipelines/adapter3d_mixin.py:68-135, the method's source taken from the file with `ast`; the module itself cannot be imported:
diffusers / mmcv are absent) over the deterministic stand-in UNet / ControlNet of tests/stubs.py.  Nothing is copied into the repo.
Run from the repo root (needs /root/reference):  python tests/golden/make_mixin_golden.py"""
import ast
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import stubs  # noqa: E402

REF = '/root/reference/lib/pipelines/adapter3d_mixin.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mixin_ref.npz')


def reference_methods():
    """get_noise_pred, get_noise_pred_p1, get_noise_pred_p2 of the reference, executed in a namespace where the third-party /
    heavyweight callees are the stand-ins of tests/stubs.py (MultiControlNetModel, unet_enc, unet_dec)."""
    from copy import copy
    tree = ast.parse(open(REF).read())
    ns = dict(torch=torch, copy=copy, MultiControlNetModel=stubs.StubMulti, unet_enc=stubs.stub_unet_enc, unet_dec=stubs.stub_unet_dec)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == 'Adapter3DMixin':
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name in ('get_noise_pred', 'get_noise_pred_p1', 'get_noise_pred_p2'):
                    exec(compile(ast.Module([fn], []), REF, 'exec'), ns)
    return ns['get_noise_pred'], ns['get_noise_pred_p1'], ns['get_noise_pred_p2']


def main():
    fn, p1, p2 = reference_methods()

    class Pipe:
        unet, controlnet = stubs.StubUNet(), stubs.StubControlNet()

    class Pipe2:
        unet = object()
        controlnet = stubs.StubMulti([stubs.StubNet(0), stubs.StubNet(1), stubs.StubNet(2)])
        negative_prompt_embeds = torch.linspace(-1, 1, 7 * 16).view(1, 7, 16)
    out = {}
    with torch.no_grad():
        for name, kw in stubs.cases().items():
            out[name] = fn(Pipe(), **kw).numpy()
        for name, kw in stubs.cases_2pass().items():
            pipe = Pipe2()
            n_extra = len(kw['p1'].get('extra_control_batches') or [])
            pipe.controlnet = stubs.StubMulti([stubs.StubNet(k) for k in range(2 + n_extra)])
            noise1, dec_args, dec_kwargs = p1(pipe, **kw['p1'])
            noise2 = p2(pipe, dec_args=dec_args, dec_kwargs=dec_kwargs, **kw['p2'])
            out[f'2pass_{name}_p1'], out[f'2pass_{name}_p2'] = noise1.numpy(), noise2.numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
