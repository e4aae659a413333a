mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null
echo "##### probe"; ./tools/probe/probe > gpurun_out/probe.log 2>&1; tail -12 gpurun_out/probe.log
export MVE_RUN_PENDING=1
echo "##### failing tests, full trace"
timeout 600 python -m pytest tests/test_lpips.py tests/test_mesh_reg.py "tests/test_mesh_loss.py::test_hip_full_size_views_and_timing" -q -m gpu -p no:cacheprovider -s --tb=short 2>&1 | grep -v "^E    \+ " > gpurun_out/retest.log; grep -E "passed|failed|Error|assert|^tests" gpurun_out/retest.log | cut -c1-300 | head -40
echo "##### mesh loop debug"
timeout 300 python tools/debug_mesh_loop.py 2>&1 | tail -40 | tee gpurun_out/debug_mesh_loop.log
echo "##### benchmark-shape parity"
timeout 900 python -m pytest tests/test_unet.py -k "benchmark_shape or full_size_vs_oracle or true_tiling" -q -m gpu -p no:cacheprovider -s --tb=short 2>&1 | tail -30 | cut -c1-400 | tee gpurun_out/parity_bench_shape.log
