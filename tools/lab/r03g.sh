#!/bin/bash
# round 3, call g: LSTM backward with data-tagged hand-off (parity first, then A/B)
OUT=gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_lstm_seq_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; RC=$?; echo "tests1 rc=$RC"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "x2, granule fwd, counter bwd: $(ms $OUT/a_$rep.json)"
  if [ $RC -eq 0 ]; then TPGSR_LSTM_GRANULE_BWD=1 timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "x2, granule fwd + bwd:        $(ms $OUT/b_$rep.json)"; fi
done
if [ $RC -eq 0 ]; then TPGSR_LSTM_GRANULE_BWD=1 timeout 150 python -m pytest tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests2.log 2>&1; echo "tests2 (crnn, granule bwd) rc=$?"; tail -3 $OUT/tests2.log; fi
