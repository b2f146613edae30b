#!/bin/bash
# round 3, call f: LSTM forward with data-tagged hand-off (parity first, then A/B); GRU prefetch depth in the step
OUT=gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_lstm_seq_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; RC=$?; echo "tests1 rc=$RC"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  TPGSR_GRU_PF=4 timeout 60 $B > $OUT/pf4_$rep.json 2> $OUT/pf4_$rep.err; echo "x2 GRU PF 4: $(ms $OUT/pf4_$rep.json)"
  TPGSR_GRU_PF=8 timeout 60 $B > $OUT/pf8_$rep.json 2> $OUT/pf8_$rep.err; echo "x2 GRU PF 8 (default), LSTM counter hand-off: $(ms $OUT/pf8_$rep.json)"
  if [ $RC -eq 0 ]; then TPGSR_LSTM_GRANULE=1 timeout 60 $B > $OUT/gran1_$rep.json 2> $OUT/gran1_$rep.err; echo "x2 LSTM granule hand-off: $(ms $OUT/gran1_$rep.json)"; fi
done
if [ $RC -eq 0 ]; then TPGSR_LSTM_GRANULE=1 timeout 150 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests2.log 2>&1; echo "tests2 (crnn, granule) rc=$?"; tail -3 $OUT/tests2.log; fi
