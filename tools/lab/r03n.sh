#!/bin/bash
# round 3, call n: full GPU suite + smoke + the bench lines and profiles that go into profiles/r03n_*
OUT=gpurun_out/r03n; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > $OUT/tests_all.log 2>&1; echo "tests_all rc=$?"; tail -4 $OUT/tests_all.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 400 python bench.py --steps 60 --warmup 15 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench default rc=$?"; tail -4 $OUT/bench_c3.err
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-traffic --alt-prec none"
for P in x3 x3b2 bf16; do timeout 120 $B --prec $P > $OUT/bench_c3_$P.json 2> $OUT/bench_c3_$P.err; echo "c3 $P rc=$?"; done
timeout 120 $B --config c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"
timeout 160 python bench.py --config c5 --steps 30 --warmup 8 --no-cpu-baseline --no-traffic --alt-prec none > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03n/bench_*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f.split("/")[-1], d["arithmetic_policy"], d["ms_per_step"], "ms", d["value"], "img/s  frac", r.get("frac"), "achieved", r.get("achieved"), "peak", r.get("peak"), "traffic", r.get("traffic"), "alt", d.get("alt_precision", {}).get("ms_per_step"), "cpu", d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
for P in x2 x3; do
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_$P -o c3 -- python $GRAFT_REPO_ROOT/bench.py --prec $P --steps 25 --warmup 5 --no-cpu-baseline --no-roofline --alt-prec none > $GRAFT_REPO_ROOT/$OUT/prof_$P.log 2>&1); echo "prof $P rc=$?"
DB=$(find $OUT/prof_$P -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c3_$P.md > /dev/null
[ -n "$DB" ] && python tools/trace_timeline.py $DB > $OUT/timeline_c3_$P.txt 2>&1
find $OUT/prof_$P -name "*.db" -size +30M -delete
done
timeout 100 python tools/lab/step_phases.py > $OUT/phases_x2.md 2>/dev/null; cat $OUT/phases_x2.md
Bq="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
timeout 60 $Bq > $OUT/p0.json 2>/dev/null; echo "x2 default priorities: $(ms $OUT/p0.json)"
TPGSR_BENCH_MAIN_PRIORITY=-1 timeout 60 $Bq > $OUT/p1.json 2>/dev/null; echo "x2 main stream high priority: $(ms $OUT/p1.json)"
