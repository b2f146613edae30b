#!/bin/bash
OUT=gpurun_out/r02j; mkdir -p $OUT
timeout 300 python -m pytest tests/test_conv_xbf_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "split or tr_read or kernel_level" > $OUT/tests_a.log 2>&1; echo "tests a rc=$?"; tail -3 $OUT/tests_a.log
for P in 64 128; do
  TPGSR_XBF_TILE=$P timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "conv or wgrad or tail" > $OUT/tests_tile$P.log 2>&1; echo "tests tile=$P rc=$?"; tail -3 $OUT/tests_tile$P.log
done
for P in 0 64 128; do
  TPGSR_XBF_TILE=$P timeout 300 python tools/bench_conv_prec.py > $OUT/conv_prec_tile$P.md 2>&1; echo "== tile $P"; grep "^|" $OUT/conv_prec_tile$P.md | cut -d'|' -f2,4,5,6
done
