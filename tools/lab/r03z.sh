#!/bin/bash
# round 3, call z: gradient exchange without the join at the end of the SR backward plan (SR bucket launched from the weight-gradient stream):
# 2-rank parity (gloo, shared GPU), RCCL at world size 1, and what the collective path costs a step at world size 1
OUT=gpurun_out/r03z; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_ddp_gpu.py tests/test_rccl_world1_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'], d['config']['gradient_exchange'])" 2>/dev/null; }
for rep in 1 2; do
  timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "plain world-1 step: $(ms $OUT/a_$rep.json)"
  timeout 90 $B --force-collectives > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "RCCL forced, SR bucket from the side stream (no join): $(ms $OUT/b_$rep.json)"
  TPGSR_DEFER_JOIN=0 timeout 90 $B --force-collectives > $OUT/c_$rep.json 2> $OUT/c_$rep.err; echo "RCCL forced, SR backward plan joins (as before): $(ms $OUT/c_$rep.json)"
done
tail -3 $OUT/b_1.err
