#!/bin/bash
# First GPU run of the tests written after round 1's GPU budget was spent (tone mapping / shading, mesh geometry gradients, LPIPS).  Each file runs in its own
# process so that a device fault in one cannot hide the others.  ~30 s of box time.  Logs land in gpurun_out/.
mkdir -p gpurun_out
export MVE_RUN_PENDING=1
for f in tests/test_tonemapping.py tests/test_mesh_grad.py tests/test_lpips.py; do
    echo "== $f"
    timeout 120 python -m pytest "$f" -q -m gpu -p no:cacheprovider -s 2>&1 | tail -25 | tee "gpurun_out/pending_$(basename "$f" .py).log" | grep -E "passed|failed|^E  |rel|engine|^FAILED" | head -20
done
