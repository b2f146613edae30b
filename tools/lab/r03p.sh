#!/bin/bash
OUT=gpurun_out/r03p; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider -k "hipgraph or dropin" > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -12 $OUT/tests1.log
timeout 100 python bench.py --steps 40 --warmup 10 --graph --no-cpu-baseline --no-roofline --alt-prec none > $OUT/graph.json 2> $OUT/graph.err; echo "graph bench rc=$?"; python -c "import json;print(json.load(open('$OUT/graph.json'))['ms_per_step'])"; tail -2 $OUT/graph.err
