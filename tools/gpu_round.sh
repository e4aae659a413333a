#!/bin/bash
# One GPU call that re-establishes the round's state: full GPU suite, smoke, bench.  Logs land in gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_round.sh'
mkdir -p gpurun_out
python -c 'import torch' 2>/dev/null
echo "##### pytest -m gpu"
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short 2>&1 | grep -v "^E    \+ " | tail -25 | cut -c1-300 | tee gpurun_out/pytest_gpu.log
echo "##### smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
echo "##### bench"
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.log | tail -1 | cut -c1-2600
