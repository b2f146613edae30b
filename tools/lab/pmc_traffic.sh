#!/bin/bash
# HBM traffic counters of one conv shape: pmc_traffic.sh OUTDIR "shape args" mode
OUT=$1; SHAPE=$2; MODE=$3
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for SET in "FETCH_SIZE WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  D=$R/$OUT/$(echo $SET | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o p -- python $R/tools/lab/one_conv.py $SHAPE $MODE > $D.log 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")[:40]
    if "conv" not in k: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v[2:]) / max(1, len(v[2:])), 1) for c, v in d.items()})
PY
done
