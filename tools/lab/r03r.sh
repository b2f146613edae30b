#!/bin/bash
# round 3, call r: more epilogue fusions (InfoGen / bn6 sums, the tail's mish backward) + side-section batching (one fork per block): parity, A/B
OUT=gpurun_out/r03r; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_bnb_fuse_gpu.py -m gpu -q -x -p no:cacheprovider -s > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $OUT/tests1.log
timeout 500 python -m pytest tests/test_tsrn_gpu.py tests/test_crnn_gpu.py tests/test_policy_x2_gpu.py tests/test_opt_student_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 $OUT/tests2.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'], d['config']['kernel_launches_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  TPGSR_BNB_FUSE=0 TPGSR_SIDE_BATCH=0 timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "x2, no epilogue fusion, a fork per side section: $(ms $OUT/a_$rep.json)"
  TPGSR_SIDE_BATCH=0 timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "x2, epilogue fusions, a fork per side section:    $(ms $OUT/b_$rep.json)"
  timeout 60 $B > $OUT/c_$rep.json 2> $OUT/c_$rep.err; echo "x2, epilogue fusions, one fork per block (default): $(ms $OUT/c_$rep.json)"
done
for cfg in c2 c5; do timeout 90 $B --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; echo "$cfg: $(ms $OUT/bench_$cfg.json)"; done
