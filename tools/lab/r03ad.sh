#!/bin/bash
# round 3, call ad: is the default step bitwise reproducible run to run -- with HIP's default four hardware queues and with eight?
OUT=gpurun_out/r03ad; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout 100 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider -k "two_independent or hipgraph" > $OUT/t4_$rep.log 2>&1; echo "4 queues, rep $rep rc=$?"; tail -1 $OUT/t4_$rep.log
  GPU_MAX_HW_QUEUES=8 timeout 100 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider -k "two_independent or hipgraph" > $OUT/t8_$rep.log 2>&1; echo "8 queues, rep $rep rc=$?"; tail -1 $OUT/t8_$rep.log
done
grep -h "AssertionError" $OUT/*.log | head -5
