#!/bin/bash
OUT=gpurun_out/r02p2; mkdir -p $OUT
timeout 150 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
timeout 100 $B > $OUT/on_$rep.json 2> $OUT/err.log; echo "early reduce: $(python -c "import json;print(json.load(open('$OUT/on_$rep.json'))['ms_per_step'])")"
TPGSR_CRNN_EARLY_REDUCE=0 timeout 100 $B > $OUT/off_$rep.json 2>> $OUT/err.log; echo "one reduce: $(python -c "import json;print(json.load(open('$OUT/off_$rep.json'))['ms_per_step'])")"
done
