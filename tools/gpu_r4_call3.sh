#!/bin/bash
# Round 4, call 3: the residual-pair mode -- kernel tests, engine parity (benchmark shape), per-op cost at 64 images next to the plain mode.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_unet_ops.py -x -q -m gpu -k "pair or chain or pingpong_main" -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r04_pytest_pair_ops.log; echo "ops rc=$?"; tail -5 gpurun_out/r04_pytest_pair_ops.log
timeout 900 python -m pytest tests/test_unet.py -x -q -m gpu -k "pair" -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r04_pytest_pair_engine.log; tail -12 gpurun_out/r04_pytest_pair_engine.log
timeout 300 python tools/op_list.py 64 > gpurun_out/r04_oplist_plain.log 2>&1; tail -1 gpurun_out/r04_oplist_plain.log
MVE_RESIDUAL_PAIR=1 timeout 300 python tools/op_list.py 64 > gpurun_out/r04_oplist_pair.log 2>&1; tail -1 gpurun_out/r04_oplist_pair.log
