#!/bin/bash
# First GPU run of the tests written after round 1's GPU budget was spent (tone mapping / shading, mesh geometry gradients, LPIPS, the
# experimental attention variant).  Each group runs in its own process so that a device fault in one cannot hide the others.
# ~40 s of box time.  Logs land in gpurun_out/.
mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null      # the first import on a fresh box pages the image in (1-2 min): keep it out of the per-group timeouts
export MVE_RUN_PENDING=1
run() {   # run <log name> <pytest args...>
    local name=$1; shift
    echo "== $name"
    timeout 300 python -m pytest "$@" -q -m gpu -p no:cacheprovider -s 2>&1 | tail -25 | tee "gpurun_out/pending_${name}.log" | grep -E "passed|failed|^E  |rel|engine|^FAILED| ms$|loop|reference kernel" | head -20
}
run tonemapping tests/test_tonemapping.py
run mesh_grad tests/test_mesh_grad.py
run lpips tests/test_lpips.py
run attention_variant tests/test_unet_ops.py -k "experimental_variant or conflict_free or vt_store_swizzle"
run graph_replay tests/test_unet.py -k graph_replay
run recon_loss tests/test_recon_loss.py
run mesh_reg tests/test_mesh_reg.py
run mesh_loss tests/test_mesh_loss.py
run blur tests/test_blur.py
run shencoder tests/test_shencoder.py
