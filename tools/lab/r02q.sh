#!/bin/bash
OUT=gpurun_out/r02q; mkdir -p $OUT
timeout 30 python bench.py --config c5 --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/err.log; echo "c5 $(python -c "import json;print(json.load(open('$OUT/bench_c5.json'))['ms_per_step'])")"
timeout 25 python bench.py --prec bf16 --steps 50 --warmup 12 --no-cpu-baseline > $OUT/bench_c3_bf16.json 2>> $OUT/err.log; echo "bf16 $(python -c "import json;print(json.load(open('$OUT/bench_c3_bf16.json'))['ms_per_step'])")"
timeout 25 python bench.py --prec f32 --steps 50 --warmup 12 --no-cpu-baseline > $OUT/bench_c3_f32.json 2>> $OUT/err.log; echo "f32 $(python -c "import json;print(json.load(open('$OUT/bench_c3_f32.json'))['ms_per_step'])")"
