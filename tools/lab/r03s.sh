#!/bin/bash
# round 3, call s: side-batch granularity (forks per backward pass): parity of the default, then an interleaved A/B
OUT=gpurun_out/r03s; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_tsrn_gpu.py tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'], d['config']['kernel_launches_per_step'])" 2>/dev/null; }
run() { tag=$1; shift; env "$@" timeout 60 $B > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag [$*]: $(ms $OUT/$tag.json)"; }
for rep in 1 2; do
  run a_$rep TPGSR_SIDE_BATCH_TAIL=0 TPGSR_SIDE_BATCH_CRNN=0
  run b_$rep TPGSR_SIDE_BATCH_TAIL=1 TPGSR_SIDE_BATCH_CRNN=0
  run c_$rep TPGSR_SIDE_BATCH_TAIL=1 TPGSR_SIDE_BATCH_CRNN=1
  run d_$rep TPGSR_SIDE_BATCH_TAIL=1 TPGSR_SIDE_BATCH_CRNN=2
  run e_$rep TPGSR_SIDE_BATCH_TAIL=1 TPGSR_SIDE_BATCH_CRNN=1 TPGSR_SIDE_BATCH_BLOCKS=2
  run f_$rep TPGSR_SIDE_BATCH_TAIL=1 TPGSR_SIDE_BATCH_CRNN=1 TPGSR_SIDE_BATCH_BLOCKS=5
done
