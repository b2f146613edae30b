#!/bin/bash
# HBM traffic of one training step from the PMC counters (separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes):
#   tools/pmc_step.sh OUTDIR [config] [steps]   ->   OUTDIR/traffic.json
OUT=$1; CFG=${2:-c3}; STEPS=${3:-4}
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/$C -o p -- python $R/tools/pmc_step_run.py --config $CFG --steps $STEPS > $R/$OUT/$C.log 2>&1
  echo "pmc $C rc=$?"
done
F=$(find $R/$OUT/FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $R/$OUT/WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/pmc_step_report.py $R/$OUT/traffic.json $F $W
# keep the summaries, drop the bulky per-dispatch CSVs
find $R/$OUT -name "*.csv" -size +8M -delete
