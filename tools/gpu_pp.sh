#!/bin/bash
# First hardware run of the ping-pong GEMM loop: one tiny launch under a short timeout (a barrier mismatch would hang), then the
# bit-identity tests, then the same-box A/B against the two-stage loop.
mkdir -p gpurun_out
timeout 120 python - <<'PY' > gpurun_out/pp_first.log 2>&1
import torch
from mvedit_amd import ops, _lib
tune = _lib.raw('mve_gemm_tune')
torch.manual_seed(0)
for (M, N, K) in [(512, 320, 64), (512, 320, 320), (4096, 640, 1280), (777, 256, 640)]:
    a = torch.randn(M, K, device='cuda', dtype=torch.float16); w = torch.randn(N, K, device='cuda', dtype=torch.float16) * K ** -0.5
    tune(1 | (1 << 27)); ref = ops.gemm(a, w)
    tune(1); out = ops.gemm(a, w); torch.cuda.synchronize()
    print(M, N, K, 'equal', torch.equal(out, ref), 'maxdiff', float((out.float() - ref.float()).abs().max()), 'nbad', int((out != ref).sum()), flush=True)
B, H, C = 2, 16, 320
x = torch.randn(B * H * H, C, device='cuda', dtype=torch.float16); w = torch.randn(320, C // 64, 3, 3, 64, device='cuda', dtype=torch.float16) * (9 * C) ** -0.5
tune(1 | (1 << 27)); ref = ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=False)[0]
tune(1); out = ops.conv3x3(x, w, B, H, H, flags=ops.W_CHUNK64, splitk=False)[0]; torch.cuda.synchronize()
print('conv equal', torch.equal(out, ref), 'maxdiff', float((out.float() - ref.float()).abs().max()), 'nbad', int((out != ref).sum()), flush=True)
PY
rc=$?; cat gpurun_out/pp_first.log; echo "first rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout 600 python -m pytest tests/test_unet_ops.py -m gpu -x -q -k "pingpong or bit_identical or big_tile or shortcut or pad_bottom or 256_wide or splitk" > gpurun_out/pp_tests.log 2>&1; tail -15 gpurun_out/pp_tests.log
timeout 600 python tools/ab_gemm_big.py 64 > gpurun_out/ab_pp.log 2>&1; cat gpurun_out/ab_pp.log
