#!/bin/bash
# round 3, call d: STN-head backward on a third (leaf) stream: bitwise tests, A/B; panel kernel restricted to K <= 96
OUT=gpurun_out/r03d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tsrn_gpu.py tests/test_conv_panel_gpu.py tests/test_fullsize_gpu.py tests/test_crnn_gpu.py tests/test_opt_student_gpu.py tests/test_next_models_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -5 $OUT/tests1.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  for P in x3 x2; do
    TPGSR_LEAF_STN=0 timeout 200 $B --prec $P > $OUT/${P}_off_$rep.json 2> $OUT/${P}_off_$rep.err; echo "$P no leaf: $(ms $OUT/${P}_off_$rep.json)"
    TPGSR_LEAF_STN=1 timeout 200 $B --prec $P > $OUT/${P}_on_$rep.json 2> $OUT/${P}_on_$rep.err; echo "$P leaf:    $(ms $OUT/${P}_on_$rep.json)"
  done
done
TPGSR_XBF_PANEL_K192=1 timeout 200 $B --prec x2 > $OUT/x2_k192.json 2> $OUT/x2_k192.err; echo "x2 leaf + panel K192: $(ms $OUT/x2_k192.json)"
timeout 200 $B --prec x2 --config c2 > $OUT/c2_x2.json 2> $OUT/c2.err; echo "c2 x2: $(ms $OUT/c2_x2.json)"
timeout 200 $B --prec x2 --config c5 --steps 30 --warmup 8 > $OUT/c5_x2.json 2> $OUT/c5.err; echo "c5 x2: $(ms $OUT/c5_x2.json)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --prec x2 --steps 25 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/prof_c3.log 2>&1); echo "prof rc=$?"
DB=$(find $OUT/prof_c3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c3.md > /dev/null
[ -n "$DB" ] && python tools/trace_timeline.py $DB > $OUT/timeline_c3.txt 2>&1
[ -n "$DB" ] && python tools/trace_timeline.py $DB 2 --dump > $OUT/timeline_c3_dump.txt 2>&1
head -32 $OUT/timeline_c3.txt
find $OUT/prof_c3 -name "*.db" -size +30M -delete
