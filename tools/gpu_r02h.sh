#!/bin/bash
OUT=gpurun_out/r02h; mkdir -p $OUT
for P in 64 128; do
  TPGSR_XBF_PIPE=$P timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "conv or wgrad or tail" > $OUT/tests_pipe$P.log 2>&1; echo "tests pipe=$P rc=$?"; tail -3 $OUT/tests_pipe$P.log
done
for P in 0 64 128; do
  TPGSR_XBF_PIPE=$P timeout 300 python tools/bench_conv_prec.py > $OUT/conv_prec_pipe$P.md 2>&1; echo "== pipe $P"; grep "^|" $OUT/conv_prec_pipe$P.md | cut -d'|' -f2,3,5,6
done
