#!/bin/bash
# round 3, call u: stride-(2,1) max-pool backward fast path: parity, A/B
OUT=gpurun_out/r03u; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'], d['config']['kernel_launches_per_step'])" 2>/dev/null; }
run() { tag=$1; shift; env "$@" timeout 60 $B > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag [$*]: $(ms $OUT/$tag.json)"; }
for rep in 1 2 3; do
  run a_$rep TPGSR_POOL2X2_FAST=0
  run b_$rep X=1
done
