#!/bin/bash
# round 3, call i: leaf section of the SR backward opened earlier (block 0's convolution gradients, block1, STN head)
OUT=gpurun_out/r03i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_tsrn_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; RC=$?; echo "tests1 rc=$RC"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  TPGSR_LEAF_EARLY=0 timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "x2, leaf = STN head only: $(ms $OUT/a_$rep.json)"
  TPGSR_LEAF_EARLY=1 timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "x2, leaf from block 0:    $(ms $OUT/b_$rep.json)"
done
