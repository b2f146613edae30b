#!/bin/bash
# round 3, call ab: does cutting the text-prior generator's backward plan in two cost the plain step anything?  the driver's command line once more
# (stdout must hold the JSON line and nothing else, also with RCCL in the process)
OUT=gpurun_out/r03ab; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
wall() { grep -o "wall [0-9.]* ms/step" $1 | tail -1; }
for rep in 1 2; do
  TPGSR_CRNN_BWD_SPLIT=0 timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "one backward plan: $(wall $OUT/a_$rep.err)"
  timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "two backward plans (default): $(wall $OUT/b_$rep.err)"
done
timeout 90 $B --force-collectives > $OUT/c.json 2> $OUT/c.err; echo "forced collectives: $(wall $OUT/c.err); stdout lines: $(wc -l < $OUT/c.json)"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c3_driver_cmd.json 2> $OUT/bench_c3_driver_cmd.err; echo "driver command rc=$? stdout lines: $(wc -l < $OUT/bench_c3_driver_cmd.json)"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03ab/bench_c3_driver_cmd.json")); r = d["roofline"]
print(d["ms_per_step"], d["value"], "frac", r["frac"], "traffic", r["traffic"], "alt", d["alt_precision"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"], "ops", d["config"]["kernel_launches_per_step"])
PY
