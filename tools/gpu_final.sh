#!/bin/bash
# End-of-round GPU call when the budget does not allow the whole suite again: the test files whose kernels changed since the last
# full run (everything that goes through the GEMM / conv kernels), smoke, bench, and the rocprofv3 passes of the bench command.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_final.sh'
mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null
echo "##### pytest -m gpu (GEMM users)"
timeout 900 python -m pytest tests/test_unet_ops.py tests/test_unet.py tests/test_vae.py tests/test_controlnet.py tests/test_pipeline_mixin.py \
    tests/test_image_enhancer.py tests/test_abi_errors.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | grep -v "^E    \+ " | tail -15 | cut -c1-300 | tee gpurun_out/pytest_gpu_final.log
echo "##### smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
echo "##### bench"
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.log | tail -1 | cut -c1-1200
echo "##### rocprofv3"
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
