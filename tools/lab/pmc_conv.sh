#!/bin/bash
# usage: pmc_conv.sh OUTDIR "shape args" mode [wgrad]   -> per-kernel counter averages
OUT=$1; SHAPE=$2; MODE=$3; WG=$4
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/$OUT/p$i -o p -- python $R/tools/lab/one_conv.py $SHAPE $MODE $WG > $R/$OUT/p$i.log 2>&1
  F=$(find $R/$OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")[:40]
    if "conv" not in k: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v[2:]) / max(1, len(v[2:]))) for c, v in d.items()})
PY
done
