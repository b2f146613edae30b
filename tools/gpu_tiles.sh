#!/bin/bash
# halo kernels: tests, then the per-shape timing table (forward and weight gradient)
OUT=gpurun_out/${1:-tiles}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_xbf_gpu.py -m gpu -q --maxfail=30 -p no:cacheprovider -s -k "halo or split or wgrad" > $OUT/tests_a.log 2>&1; echo "tests a rc=$?"; tail -2 $OUT/tests_a.log; grep -E "^halo|^FAILED|^ERROR|^E  " $OUT/tests_a.log | head -50
timeout 300 python tools/bench_conv_prec.py > $OUT/conv_prec.md 2>&1; grep "^|" $OUT/conv_prec.md | cut -d'|' -f2,5,6,8,9; grep -i "error\|Traceback" -A5 $OUT/conv_prec.md | head -20
