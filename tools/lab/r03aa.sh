#!/bin/bash
# round 3, call aa: the text-prior generator's backward pass as two plans with its early gradient bucket launched in between: parity (one rank,
# two ranks over gloo, RCCL at world size 1), and the step with / without the collectives
OUT=gpurun_out/r03aa; mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_crnn_gpu.py tests/test_ddp_gpu.py tests/test_rccl_world1_gpu.py tests/test_policy_x2_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
wall() { grep -o "wall [0-9.]* ms/step" $1 | tail -1; }
for rep in 1 2; do
  timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "plain: $(wall $OUT/a_$rep.err)"
  timeout 90 $B --force-collectives > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "RCCL forced, three buckets: $(wall $OUT/b_$rep.err)"
done
python -c "import json;print(json.load(open('$OUT/b_1.json'))['config']['gradient_exchange'])"
