"""Writes tests/golden/vae_tiny.npz: the VAE oracle (oracle/vae_oracle.py) on seeded miniature configurations, so that a change
to the restatement is noticed.  (diffusers is absent: this pins the oracle against itself only -- see the oracle's header.)
Run from the repo root:  python tests/golden/make_vae_golden.py"""
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import vae_oracle as V  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'vae_tiny.npz')


def cases():
    out = {}
    for tag, cfg, seed in (('tiny', V.TINY_VAE, 1), ('odd', V.ODD_VAE, 2)):
        sd = V.random_params(cfg, seed)
        g = torch.Generator().manual_seed(seed + 10)
        z = torch.randn(2, 4, 8, 8, generator=g)
        x = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
        with torch.no_grad():
            out[f'{tag}_decode'] = V.decode(sd, cfg, z).numpy()
            out[f'{tag}_moments'] = V.encode_moments(sd, cfg, x).numpy()
    return out


if __name__ == '__main__':
    o = cases()
    np.savez_compressed(OUT, **o)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes', {k: v.shape for k, v in o.items()})
