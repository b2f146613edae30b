#!/bin/bash
# round 3, call t: teacher forward launched after the student's (fills the SR network's BiGRU kernels), side batch split at gru2: parity, A/B
# (TPGSR_TEACHER_LATE was removed from the tree afterwards: slower, and its bitwise test differed once in three suite runs)
OUT=gpurun_out/r03t; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider -k "teacher_late or hipgraph or golden" > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'], d['config']['kernel_launches_per_step'])" 2>/dev/null; }
run() { tag=$1; shift; env "$@" timeout 60 $B > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag [$*]: $(ms $OUT/$tag.json)"; }
for rep in 1 2; do
  run a_$rep X=1
  run b_$rep TPGSR_TEACHER_LATE=1
  run c_$rep TPGSR_SIDE_BATCH_SPLIT=1
  run d_$rep TPGSR_SIDE_BATCH_SPLIT=1 TPGSR_SIDE_BATCH_BLOCKS=1
  run e_$rep TPGSR_TEACHER_LATE=1 TPGSR_SIDE_BATCH_SPLIT=1
done
