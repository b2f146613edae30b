#!/bin/bash
# scheduling levers on the C3 step: weight-gradient stream priority / CU mask, hipGraph replay
OUT=gpurun_out/r02w; mkdir -p $OUT
python -c "import torch; print('priority range (least, greatest):', torch.cuda.Stream.priority_range())" 2>&1 | tail -1
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
run() { local tag=$1; shift; env "$@" timeout 200 $B $EXTRA > $OUT/$tag.json 2>> $OUT/err.log; echo "$tag: $(python -c "import json;d=json.load(open('$OUT/$tag.json'));print(d['ms_per_step'])" 2>&1 | tail -1)"; }
for rep in 1 2; do
run default_$rep A=1
run side_low_$rep TPGSR_SIDE_PRIORITY=1
run side_high_$rep TPGSR_SIDE_PRIORITY=-1
run mask192_$rep TPGSR_SIDE_CUMASK=192
run mask128_$rep TPGSR_SIDE_CUMASK=128
done
EXTRA=--graph run graph_1 A=1
EXTRA=--graph run graph_2 A=1
tail -3 $OUT/err.log
