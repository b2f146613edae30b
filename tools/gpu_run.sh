#!/bin/bash
# One gpurun call = tests + bench lines + kernel-trace profile (outputs under gpurun_out/$TAG, merged back by gpurun).
# usage: tools/gpu_run.sh TAG [what...]   what in: tests bench_c3 bench_c2 bench_c5 prof_c3 prof_c2 smoke
TAG=${1:-run}; shift
WHAT=${@:-tests bench_c3 bench_c2 prof_c3 smoke}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in $WHAT; do
  case $w in
    tests)    timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -x -p no:cacheprovider ${PYTEST_ARGS} > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/tests.log ;;
    tests_all) timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider -s ${PYTEST_ARGS} > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/tests.log ;;
    bench_c3) timeout 600 python bench.py --steps 60 --warmup 15 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench_c3 rc=$?" | tee -a $OUT/summary.txt; cat $OUT/bench_c3.json ;;
    bench_c2) timeout 600 python bench.py --config c2 --steps 60 --warmup 15 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench_c2 rc=$?" | tee -a $OUT/summary.txt; cat $OUT/bench_c2.json ;;
    bench_c5) timeout 600 python bench.py --config c5 --steps 30 --warmup 8 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench_c5 rc=$?" | tee -a $OUT/summary.txt; cat $OUT/bench_c5.json ;;
    prof_c3)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof_c3.log 2>&1); echo "prof_c3 rc=$?" | tee -a $OUT/summary.txt
              DB=$(find $OUT/prof_c3 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c3.md > /dev/null; [ -n "$DB" ] && python tools/trace_timeline.py $DB > $OUT/timeline_c3.txt 2>&1; head -30 $OUT/kernel_stats_c3.md; find $OUT/prof_c3 -name "*.db" -size +30M -delete ;;
    prof_c2)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps 25 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof_c2.log 2>&1); echo "prof_c2 rc=$?" | tee -a $OUT/summary.txt
              DB=$(find $OUT/prof_c2 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c2.md > /dev/null; head -20 $OUT/kernel_stats_c2.md; find $OUT/prof_c2 -name "*.db" -size +30M -delete ;;
    smoke)    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.log ;;
    *)        timeout 900 bash -c "$w" > $OUT/custom.log 2>&1; echo "custom rc=$?" | tee -a $OUT/summary.txt; tail -30 $OUT/custom.log ;;
  esac
done
