#!/bin/bash
# One gpurun call = tests + bench lines + kernel-trace profile (outputs under gpurun_out/$TAG, merged back by gpurun).
# usage: tools/gpu_run.sh TAG [what...]   what in: tests tests_all bench_c3 bench_c2 bench_c5 bench_eval bench_opt ab_c3 plan_gaps prof_c3 prof_c2
#        smoke, t:FILE,FILE,... (pytest on the named test files), x:CMD (a shell command, '+' for spaces: x:python+tools/lab/one_conv.py;
#        stdin is /dev/null and the limit X_TIMEOUT, default 300 s -- a bare `python` once sat on stdin for the whole 900 s of a call).
# Environment: PYTEST_ARGS (extra pytest flags), BENCH_ARGS (extra bench.py flags, e.g. "--no-traffic --alt-prec none"),
#              AB_ENV ("NAME=a NAME=b ...": ab_c3 runs the C3 bench once per setting, interleaved on this one box).
# This replaces the per-experiment scripts of rounds 2-4 (tools/lab/r0*.sh, no longer tracked): an experiment is one line of these words.
TAG=${1:-run}; shift
WHAT=${@:-tests bench_c3 bench_c2 prof_c3 smoke}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
# a stale library (a failed build leaves the old .so in place) wastes the whole call: refuse to run on one
python -c "from tpgsr_amd import build as b; assert open(b.LIB + '.stamp').read() == b._digest(), 'STALE LIBRARY: rebuild (python -m tpgsr_amd.build)'" || exit 1
summ() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d.get("roofline") or {}
print(d["ms_per_step"], "ms/step", d["value"], d["unit"], "| family", r.get("ms_per_step_replayed"), "ms frac", r.get("frac"), "traffic", r.get("traffic"),
      {k: round(v["ms"], 3) for k, v in (r.get("by_kind") or {}).items()}, "| eval", (d.get("eval") or {}).get("value"), "| opt", (d.get("tpg_opt") or {}).get("ms_per_step"))
for x in (r.get("per_shape") or [])[:16]: print("   ", x)
PY
}
for w in $WHAT; do
  case $w in
    tests)    timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -x -p no:cacheprovider ${PYTEST_ARGS} > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/tests.log ;;
    tests_all) timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider -s ${PYTEST_ARGS} > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/tests.log ;;
    t:*)      timeout 1500 python -m pytest $(echo ${w#t:} | tr "," " ") -m gpu -q -p no:cacheprovider ${PYTEST_ARGS} > $OUT/tests_sel.log 2>&1; echo "selected tests rc=$?" | tee -a $OUT/summary.txt; tail -8 $OUT/tests_sel.log ;;
    bench_eval) timeout 600 python bench.py --eval --steps 60 --warmup 15 ${BENCH_ARGS} > $OUT/bench_eval.json 2> $OUT/bench_eval.err; echo "bench_eval rc=$?" | tee -a $OUT/summary.txt; summ $OUT/bench_eval.json ;;
    bench_opt) timeout 600 python bench.py --tpg opt --steps 30 --warmup 10 --no-cpu-baseline ${BENCH_ARGS} > $OUT/bench_opt.json 2> $OUT/bench_opt.err; echo "bench_opt rc=$?" | tee -a $OUT/summary.txt; summ $OUT/bench_opt.json ;;
    ab_c3)    for kv in ${AB_ENV:-_=_}; do env $kv timeout 400 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none ${BENCH_ARGS} > $OUT/bench_c3_$kv.json 2> $OUT/bench_c3_$kv.err; echo "== $kv"; summ $OUT/bench_c3_$kv.json | head -${AB_LINES:-6}; done ;;
    plan_gaps) timeout 400 python tools/plan_gaps.py --out $OUT/plan_gaps_c3_x2.md > $OUT/plan_gaps.log 2>&1; echo "plan_gaps rc=$?" | tee -a $OUT/summary.txt; head -12 $OUT/plan_gaps_c3_x2.md ;;
    bench_c3) timeout 600 python bench.py --steps 60 --warmup 15 ${BENCH_ARGS} > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench_c3 rc=$?" | tee -a $OUT/summary.txt; summ $OUT/bench_c3.json ;;
    bench_c2) timeout 600 python bench.py --config c2 --steps 60 --warmup 15 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench_c2 rc=$?" | tee -a $OUT/summary.txt; cat $OUT/bench_c2.json ;;
    bench_c5) timeout 600 python bench.py --config c5 --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench_c5 rc=$?" | tee -a $OUT/summary.txt; cat $OUT/bench_c5.json ;;
    prof_c3)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-roofline --alt-prec none --no-module-api > $GRAFT_REPO_ROOT/$OUT/prof_c3.log 2>&1); echo "prof_c3 rc=$?" | tee -a $OUT/summary.txt
              DB=$(find $OUT/prof_c3 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c3.md > /dev/null; [ -n "$DB" ] && python tools/trace_timeline.py $DB > $OUT/timeline_c3.txt 2>&1; head -30 $OUT/kernel_stats_c3.md; find $OUT/prof_c3 -name "*.db" -size +30M -delete ;;
    prof_c2)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps 25 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof_c2.log 2>&1); echo "prof_c2 rc=$?" | tee -a $OUT/summary.txt
              DB=$(find $OUT/prof_c2 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c2.md > /dev/null; head -20 $OUT/kernel_stats_c2.md; find $OUT/prof_c2 -name "*.db" -size +30M -delete ;;
    smoke)    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.log ;;
    x:*)      timeout ${X_TIMEOUT:-300} bash -c "$(echo ${w#x:} | tr '+' ' ')" < /dev/null > $OUT/custom.log 2>&1; echo "custom rc=$?" | tee -a $OUT/summary.txt; tail -40 $OUT/custom.log ;;
    *)        echo "unknown word '$w' (a shell command goes in as x:CMD with '+' for spaces: the words of \$@ are split)" | tee -a $OUT/summary.txt ;;
  esac
done
