#!/bin/bash
# round 3, call o: non-overlapping 2x2 max-pool backward fast path: parity, A/B
OUT=gpurun_out/r03o; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -2 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  TPGSR_POOL2X2_FAST=0 timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "x2, gather-form pool backward: $(ms $OUT/a_$rep.json)"
  timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "x2, 2x2 fast path:             $(ms $OUT/b_$rep.json)"
done
