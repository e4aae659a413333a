"""Writes tests/golden/srvgg_ref.npz by EXECUTING the reference's SRVGGNetCompact (lib/models/decoders/image_space_ss.py) in this
container: the class definition is taken from the file with `ast` (the mmgen registry decorator and the mmcv import -- neither is
used by forward -- are dropped), instantiated with seeded weights, and run on seeded inputs.  Nothing is copied into the repo.
Run from the repo root (needs /root/reference):  python tests/golden/make_srvgg_golden.py"""
import ast
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import srvgg_oracle as S  # noqa: E402

REF = '/root/reference/lib/models/decoders/image_space_ss.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'srvgg_ref.npz')


def reference_class():
    tree = ast.parse(open(REF).read())
    ns = dict(nn=nn, F=F, torch=torch)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == 'SRVGGNetCompact':
            node.decorator_list = []
            exec(compile(ast.Module([node], []), REF, 'exec'), ns)
    return ns['SRVGGNetCompact']


def main():
    cls = reference_class()
    out = {}
    for tag, kw, hw in (('small', dict(num_feat=64, num_conv=3, upscale=4), (12, 20)), ('odd', dict(num_feat=32, num_conv=2, upscale=2), (9, 16))):
        sd = S.random_params(seed=7, **kw)
        net = cls(num_in_ch=3, num_out_ch=3, act_type='prelu', **kw).eval()
        net.load_state_dict(sd)
        x = torch.rand(2, 3, *hw, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            out[f'{tag}_x'] = x.numpy()
            out[f'{tag}_y'] = net(x).numpy()
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
