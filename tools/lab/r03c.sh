#!/bin/bash
# round 3, call c: row-panel kernel for the 1x1 convolutions: parity, network-level tests, A/B on the C3 step, per-shape table
OUT=gpurun_out/r03c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_panel_gpu.py tests/test_tsrn_gpu.py tests/test_blocks_gpu.py -m gpu -q -x -p no:cacheprovider -s > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; grep -E "^panel|passed|failed|Error|error" $OUT/tests1.log | tail -25
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  for P in x3 x2; do
    TPGSR_XBF_PANEL=0 timeout 200 $B --prec $P > $OUT/${P}_off_$rep.json 2> $OUT/${P}_off_$rep.err; echo "$P tile loop: $(ms $OUT/${P}_off_$rep.json)"
    TPGSR_XBF_PANEL=1 timeout 200 $B --prec $P > $OUT/${P}_on_$rep.json 2> $OUT/${P}_on_$rep.err; echo "$P panel:     $(ms $OUT/${P}_on_$rep.json)"
  done
done
for P in x3 x2; do
timeout 600 python bench.py --steps 40 --warmup 10 --prec $P --no-cpu-baseline --no-traffic > $OUT/bench_$P.json 2> $OUT/bench_$P.err; echo "bench $P rc=$?"; python -c "
import json; d=json.load(open('$OUT/bench_$P.json')); r=d['roofline']
print(d['ms_per_step'], d['value'], 'frac', r['frac'], 'achieved', r['achieved'], 'peak', r['peak'])
for k,v in r['by_kind'].items(): print(k, v['launches'], round(v['ms'],3), 'ms', round(v['tflops'],1), 'TF frac', round(v['frac'],3))
for row in r['per_shape']: print(row)
"
done
