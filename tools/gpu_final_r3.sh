#!/bin/bash
# End-of-round GPU call: the whole GPU suite, smoke, the bench line, and the rocprofv3 passes of the bench command.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/gpu_final_r3.sh'
mkdir -p gpurun_out
echo "##### pytest -m gpu (whole suite)"
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short > gpurun_out/r03_pytest_gpu_final.log 2>&1
tail -5 gpurun_out/r03_pytest_gpu_final.log | cut -c1-300
echo "##### smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "##### bench"
timeout 900 python bench.py 2>gpurun_out/r03_bench_final.err > gpurun_out/r03_bench_final.log
tail -1 gpurun_out/r03_bench_final.log | cut -c1-1500
echo "##### rocprofv3"
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
