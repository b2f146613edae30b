#!/bin/bash
# round 3, call w: the bench lines and profiles that go into profiles/r03w_* (state after the epilogue fusions / side batching / pool fast paths)
OUT=gpurun_out/r03w; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c3_driver_cmd.json 2> $OUT/bench_c3_driver_cmd.err; echo "bench (driver command) rc=$?"; tail -3 $OUT/bench_c3_driver_cmd.err
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-traffic --alt-prec none"
for P in x2 x3 x3b2 bf16; do timeout 120 $B --prec $P > $OUT/bench_c3_$P.json 2> $OUT/bench_c3_$P.err; echo "c3 $P rc=$?"; done
timeout 120 $B --config c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"
timeout 160 python bench.py --config c5 --steps 30 --warmup 8 --no-cpu-baseline --no-traffic --alt-prec none > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03w/bench_*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f.split("/")[-1], d["arithmetic_policy"], d["ms_per_step"], "ms", d["value"], "img/s  frac", r.get("frac"), "achieved", r.get("achieved"), "peak", r.get("peak"), "traffic", r.get("traffic"), "alt", d.get("alt_precision", {}).get("ms_per_step"), "cpu", d.get("cpu_baseline", {}).get("value"), "ops", d["config"]["kernel_launches_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
for P in x2; do
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_$P -o c3 -- python $GRAFT_REPO_ROOT/bench.py --prec $P --steps 25 --warmup 5 --no-cpu-baseline --no-roofline --alt-prec none > $GRAFT_REPO_ROOT/$OUT/prof_$P.log 2>&1); echo "prof $P rc=$?"
DB=$(find $OUT/prof_$P -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c3_$P.md > /dev/null
[ -n "$DB" ] && python tools/trace_timeline.py $DB > $OUT/timeline_c3_$P.txt 2>&1
done
timeout 100 python tools/lab/step_phases.py > $OUT/phases_x2.md 2>/dev/null; cat $OUT/phases_x2.md
timeout 200 python tools/lab/shape_table.py > $OUT/shape_table.json 2> $OUT/shape_table.err; echo "shape table rc=$?"; cut -c1-300 $OUT/shape_table.json
