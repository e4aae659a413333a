#!/bin/bash
# Full GPU round: build, the whole `-m gpu` suite, smoke(), one bench line.  Logs land in gpurun_out/.
mkdir -p gpurun_out
python -m mvedit_amd.build > gpurun_out/build.log 2>&1 || tail -5 gpurun_out/build.log
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_gpu.log | grep -v "where\|and  " | head -30
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.log | tail -2
