#!/bin/bash
# round 3, call h: SR forward prologue (pack, STN head, block1) on the side stream next to the student's forward pass
OUT=gpurun_out/r03h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_tsrn_gpu.py tests/test_crnn_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; RC=$?; echo "tests1 rc=$RC"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  TPGSR_SR_PRE_SIDE=0 timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "x2, SR prologue on main: $(ms $OUT/a_$rep.json)"
  TPGSR_SR_PRE_SIDE=1 timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "x2, SR prologue on side: $(ms $OUT/b_$rep.json)"
done
TPGSR_SR_PRE_SIDE=1 timeout 60 $B --config c5 --steps 30 --warmup 8 > $OUT/c5.json 2> $OUT/c5.err; echo "c5 x2: $(ms $OUT/c5.json)"
