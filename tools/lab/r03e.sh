#!/bin/bash
# round 3, call e: BiGRU prefetch rings (forward ring new, depth sweep alone and in the C3 step); LSTM forward with data-tagged hand-off
OUT=gpurun_out/r03e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_lstm_seq_gpu.py -m gpu -q -x -p no:cacheprovider -k "gru or GRU or rrb or block or lstm" > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $OUT/tests1.log
timeout 300 python tools/lab/gru_pf.py > $OUT/gru_pf.md 2>&1; cat $OUT/gru_pf.md
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  for PF in 4 8 12; do
    TPGSR_GRU_PF=$PF timeout 200 $B > $OUT/pf${PF}_$rep.json 2> $OUT/pf${PF}_$rep.err; echo "x2 GRU PF $PF: $(ms $OUT/pf${PF}_$rep.json)"
  done
  TPGSR_LSTM_GRANULE=0 timeout 200 $B > $OUT/gran0_$rep.json 2> $OUT/gran0_$rep.err; echo "x2 LSTM counter hand-off: $(ms $OUT/gran0_$rep.json)"
  TPGSR_LSTM_GRANULE=1 timeout 200 $B > $OUT/gran1_$rep.json 2> $OUT/gran1_$rep.err; echo "x2 LSTM granule hand-off: $(ms $OUT/gran1_$rep.json)"
done
timeout 900 python -m pytest tests/test_crnn_gpu.py tests/test_tsrn_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 $OUT/tests2.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-roofline --alt-prec none > $GRAFT_REPO_ROOT/$OUT/prof_c3.log 2>&1); echo "prof rc=$?"
DB=$(find $OUT/prof_c3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c3.md > /dev/null
[ -n "$DB" ] && python tools/trace_timeline.py $DB > $OUT/timeline_c3.txt 2>&1
grep -E "lstm_seq|bigru" $OUT/kernel_stats_c3.md
find $OUT/prof_c3 -name "*.db" -size +30M -delete
