#!/bin/bash
# halo kernel: kernel tests, per-shape timing table, optional PMC counters of two shapes
OUT=gpurun_out/${1:-tiles}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv_xbf_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -s -k "halo or split or tr_read or kernel_level" > $OUT/tests_a.log 2>&1; echo "tests a rc=$?"; tail -2 $OUT/tests_a.log; grep -E "^halo|^FAILED|^ERROR" $OUT/tests_a.log | head -30
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "conv or wgrad or tail" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log; grep -E "^FAILED|^ERROR" $OUT/tests.log | head -10
TPGSR_XBF_DEBUG=1 timeout 300 python tools/bench_conv_prec.py > $OUT/conv_prec.md 2>&1; grep "^|" $OUT/conv_prec.md | cut -d'|' -f2,4,5,6; grep -i "error\|Traceback" -A5 $OUT/conv_prec.md | head -20; grep "^\[tpgsr\]" $OUT/conv_prec.md | sort | uniq | head -20
if [ -n "$PMC" ]; then
  bash tools/lab/pmc_conv.sh $OUT/pmc_trunk "48 16 64 64 64 3 3 1" x3 2>&1 | tail -8
  bash tools/lab/pmc_conv.sh $OUT/pmc_conv5 "48 4 26 512 512 3 3 1" x3 2>&1 | tail -8
fi
