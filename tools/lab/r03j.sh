#!/bin/bash
# round 3, call j: kernel trace + timeline of the C3 step at the current defaults (x2)
OUT=gpurun_out/r03j; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-roofline --alt-prec none > $GRAFT_REPO_ROOT/$OUT/prof_c3.log 2>&1); echo "prof rc=$?"
DB=$(find $OUT/prof_c3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/kernel_stats_c3.md > /dev/null
[ -n "$DB" ] && python tools/trace_timeline.py $DB > $OUT/timeline_c3.txt 2>&1
[ -n "$DB" ] && python tools/trace_timeline.py $DB 2 --dump > $OUT/timeline_c3_dump.txt 2>&1
head -60 $OUT/timeline_c3.txt
find $OUT/prof_c3 -name "*.db" -size +30M -delete
