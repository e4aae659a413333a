"""Regenerates tests/golden/raymarching_small.npz from the C oracle.

The oracle itself is pinned against the reference's own kernels (oracle/_ref, built by
oracle/build_ref.py from /root/reference) in tests/test_raymarching_ref.py on the GPU box;
tests/golden/raymarching_ref_gfx950.npz holds outputs of those reference kernels.
Run from the repo root:  python tests/golden/make_raymarching_golden.py
"""
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_raymarching import _golden_outputs, GOLDEN  # noqa: E402

np.savez_compressed(GOLDEN, **_golden_outputs())
print('wrote', GOLDEN, os.path.getsize(GOLDEN), 'bytes')
