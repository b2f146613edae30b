#!/bin/bash
# round 3, call m: teacher forward two-term under the x2 policy: gates on 3 seeds, A/B
OUT=gpurun_out/r03m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_policy_x2_gpu.py -m gpu -q -x -p no:cacheprovider -s -k "x2" > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; grep -E "policy \(seed|passed|failed" $OUT/tests1.log | tail -8
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "x2, teacher two-term: $(ms $OUT/b_$rep.json)"
done
