#!/bin/bash
O=gpurun_out/r04r; mkdir -p $O
python -c "from tpgsr_amd import build as b; assert open(b.LIB+\".stamp\").read()==b._digest(), \"STALE LIBRARY\"" || exit 1
export GPU_MAX_HW_QUEUES=8
timeout 120 python tools/lab/bnb_epilogue_time.py 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_gru_gate_math_gpu.py tests/test_kernels_gpu.py tests/test_conv_halo3_gpu.py tests/test_bnb_fuse_gpu.py tests/test_gru_wgrad_gpu.py tests/test_blocks_gpu.py tests/test_tsrn_gpu.py tests/test_policy_x2_gates_gpu.py -q -m gpu 2>&1 | tail -12
