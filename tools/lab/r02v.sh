#!/bin/bash
OUT=gpurun_out/r02v; mkdir -p $OUT
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
run() { local tag=$1; shift; env "$@" timeout 200 $B $EXTRA > $OUT/$tag.json 2>> $OUT/err.log; echo "$tag: $(python -c "import json;d=json.load(open('$OUT/$tag.json'));print(d['ms_per_step'])" 2>&1 | tail -1)"; }
for rep in 1 2 3; do
run default_$rep A=1
run main_high_$rep TPGSR_BENCH_MAIN_PRIORITY=-1
run main_normal_$rep TPGSR_BENCH_MAIN_PRIORITY=0
done
tail -3 $OUT/err.log
