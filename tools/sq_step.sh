#!/bin/bash
# SQ-counter profile of every kernel of one training step: where do the family's waves wait -- memory, LDS, the matrix
# pipe's dependencies?  Two `rocprofv3 --pmc` passes (--kernel-trace only, as MI355X_MICROARCH.md prescribes; 8 SQ + 2 GRBM slots per pass):
#   tools/sq_step.sh OUTDIR [config] [steps]   ->   OUTDIR/sq_family.md (+ sq_family.json)
OUT=$1; CFG=${2:-c3}; STEPS=${3:-3}
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
i=0
for SET in "$A" "$B"; do
  i=$((i + 1))
  timeout 420 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/$OUT/pass$i -o p -- python $R/tools/pmc_step_run.py --config $CFG --steps $STEPS > $R/$OUT/pass$i.log 2>&1
  echo "sq pass $i rc=$?"
done
CSVS=$(find $R/$OUT/pass1 $R/$OUT/pass2 -name "*counter_collection.csv" | tr "\n" " ")
python $R/tools/sq_step_report.py $R/$OUT/sq_family $CSVS
find $R/$OUT -name "*.csv" -size +8M -delete
