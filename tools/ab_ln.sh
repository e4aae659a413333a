timeout 900 python -m pytest tests/test_gemm_ln.py tests/test_unet_ops.py -x -q -k "gemm_ln or layernorm or epilogue or norms_read" 2>&1 | tail -2
timeout 300 python tools/op_list.py 64 > gpurun_out/oplist64_ln1.log 2>&1
MVE_GEMM_LN_FUSE=0 timeout 300 python tools/op_list.py 64 > gpurun_out/oplist64_ln0.log 2>&1
tail -n 1 gpurun_out/oplist64_ln1.log; tail -n 1 gpurun_out/oplist64_ln0.log
