#!/bin/bash
timeout 900 python -m pytest tests/test_mesh_ops.py tests/test_bake_ref.py tests/test_mesh_forward_ref.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --secondary-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)['secondary']; print({k: d[k] for k in ('bake_multiview', 'mesh_forward', 'tracer_b7_masks')})"
