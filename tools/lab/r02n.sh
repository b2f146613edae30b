#!/bin/bash
OUT=gpurun_out/r02n; mkdir -p $OUT
timeout 200 python -m pytest tests/test_moran_gpu.py tests/test_kernels_gpu.py tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider -k "moran or wgrad or conv_fwd_plain or affine or bn or c3_step or upsample" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
