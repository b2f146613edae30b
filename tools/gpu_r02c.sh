#!/bin/bash
OUT=gpurun_out/r02c; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_conv_xbf_gpu.py -m gpu -q -s --maxfail=60 -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.log
for p in x3 bf16; do
  TPGSR_CONV_PREC=$p timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/bench_c3_$p.json 2> $OUT/bench_c3_$p.err; echo "bench c3 $p rc=$?"; cut -c1-400 $OUT/bench_c3_$p.json
done
TPGSR_CONV_PREC=x3 timeout 300 python bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline > $OUT/bench_c2_x3.json 2> $OUT/bench_c2_x3.err; echo "bench c2 x3 rc=$?"; cut -c1-300 $OUT/bench_c2_x3.json
