"""Regenerates tests/golden/unet_tiny.npz from the torch fp32 oracle (oracle/unet_oracle.py).

PARITY UNPINNED: diffusers is not importable in this environment, so these vectors pin the oracle
against itself only (see the header of oracle/unet_oracle.py).
Run from the repo root:  python tests/golden/make_unet_golden.py
"""
import os
import sys
sys.dont_write_bytecode = True      # never write __pycache__ into the read-only reference tree

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_unet import _golden_case, GOLDEN  # noqa: E402

np.savez_compressed(GOLDEN, **_golden_case())
print('wrote', GOLDEN, os.path.getsize(GOLDEN), 'bytes')
