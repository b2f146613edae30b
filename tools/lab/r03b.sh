#!/bin/bash
# round 3, call b: two-term policies (gates on 3 seeds, wgrad error), policy A/B on the C3 step, the new bench line (family roofline + PMC traffic)
OUT=gpurun_out/r03b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_policy_x2_gpu.py -m gpu -q -p no:cacheprovider -s > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; grep -E "policy \(seed|rms error|passed|failed|Error" $OUT/tests1.log | tail -20
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  for P in x3 x3b2 x2 bf16; do
    timeout 200 $B --prec $P > $OUT/${P}_$rep.json 2> $OUT/${P}_$rep.err; echo "$P: $(ms $OUT/${P}_$rep.json)"
  done
done
timeout 600 python bench.py --steps 60 --warmup 15 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench rc=$?"; tail -3 $OUT/bench_c3.err; python -c "
import json; d=json.load(open('$OUT/bench_c3.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'achieved', r['achieved'], 'peak', r['peak'], 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'])
print(json.dumps(r['by_kind'], indent=0))
for row in r['per_shape']: print(row)
print(d['step_roofline'])
"
timeout 600 python bench.py --steps 60 --warmup 15 --prec x3b2 --no-cpu-baseline --no-traffic > $OUT/bench_c3_x3b2.json 2> $OUT/bench_c3_x3b2.err; echo "bench x3b2 rc=$?"; python -c "
import json; d=json.load(open('$OUT/bench_c3_x3b2.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'achieved', r['achieved'], 'peak', r['peak'])
print(json.dumps(r['by_kind'], indent=0))
"
