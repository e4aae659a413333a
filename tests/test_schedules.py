"""Outer-loop schedules (mvedit_amd/pipelines/schedules.py) against the reference's own module-level functions
(lib/pipelines/mvedit_3d_pipeline.py:41-78) executed on a progress grid; committed golden for boxes without /root/reference
(`python tests/test_schedules.py` rewrites it).  Bit-equal: same float expressions."""
import ast
import os

import numpy as np

from mvedit_amd.pipelines import schedules as S

REF = '/root/reference/lib/pipelines/mvedit_3d_pipeline.py'
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'schedules_ref.npz')
NAMES = ['default_lr_multiplier', 'default_max_num_views', 'default_render_size_p', 'default_lr_schedule', 'default_patch_rgb_weight',
         'default_patch_normal_weight', 'default_entropy_weight', 'default_normal_reg_weight']
GRID = [i / 40 for i in range(41)] + [0.3, 0.6, 0.300001, 0.600001]


def _table(ns):
    out = {}
    for n in NAMES:
        two = n in ('default_lr_multiplier', 'default_max_num_views')
        out[n] = np.asarray([[ns[n](p, d) for p in GRID] for d in (0.5, 0.7, 0.75)] if two else [ns[n](p) for p in GRID], np.float64)
    return out


def _reference():
    ns = {}
    for node in ast.parse(open(REF).read()).body:
        if isinstance(node, ast.FunctionDef) and node.name in NAMES:
            exec(compile(ast.Module([node], []), REF, 'exec'), ns)
    return _table(ns)


def test_schedules_equal_reference_functions():
    ref = _reference() if os.path.exists(REF) else dict(np.load(GOLD))
    got = _table({n: getattr(S, n) for n in NAMES})
    for n in NAMES:
        assert np.array_equal(got[n], ref[n]), n


if __name__ == '__main__':
    np.savez_compressed(GOLD, **_reference())
    print('wrote', GOLD)
