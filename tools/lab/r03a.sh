#!/bin/bash
# round 3, call a: persistent BiLSTM kernels (parity + A/B), deferred SR-backward join, RCCL at world 1, kernel trace, per-step PMC traffic
OUT=gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lstm_seq_gpu.py tests/test_crnn_gpu.py tests/test_rccl_world1_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -15 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  TPGSR_LSTM_SEQ=0 TPGSR_DEFER_JOIN=0 timeout 200 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "old (steps, join):        $(ms $OUT/a_$rep.json) $(grep 'host submission' $OUT/a_$rep.err | sed 's/.*\] //')"
  TPGSR_LSTM_SEQ=0 TPGSR_DEFER_JOIN=1 timeout 200 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "steps, deferred join:     $(ms $OUT/b_$rep.json) $(grep 'host submission' $OUT/b_$rep.err | sed 's/.*\] //')"
  TPGSR_LSTM_SEQ=1 TPGSR_LSTM_SEQ_BWD=0 timeout 200 $B > $OUT/c_$rep.json 2> $OUT/c_$rep.err; echo "seq fwd only:             $(ms $OUT/c_$rep.json) $(grep 'host submission' $OUT/c_$rep.err | sed 's/.*\] //')"
  TPGSR_LSTM_SEQ=1 timeout 200 $B > $OUT/d_$rep.json 2> $OUT/d_$rep.err; echo "seq fwd+bwd (default):    $(ms $OUT/d_$rep.json) $(grep 'host submission' $OUT/d_$rep.err | sed 's/.*\] //')"
done
timeout 400 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests2.log 2>&1; echo "tests2 (fullsize) rc=$?"; tail -5 $OUT/tests2.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof_c3.log 2>&1); echo "prof rc=$?"
DB=$(find $OUT/prof_c3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c3.md > /dev/null
[ -n "$DB" ] && python tools/trace_timeline.py $DB > $OUT/timeline_c3.txt 2>&1
[ -n "$DB" ] && python tools/trace_timeline.py $DB 2 --dump > $OUT/timeline_c3_dump.txt 2>&1
head -40 $OUT/kernel_stats_c3.md; head -30 $OUT/timeline_c3.txt
find $OUT/prof_c3 -name "*.db" -size +30M -delete
bash tools/pmc_step.sh $OUT/pmc c3 4 2>&1 | tail -40
