#!/bin/bash
# round 3, call z2: the collective path at world size 1 with more hardware queues (GPU_MAX_HW_QUEUES: HIP multiplexes streams onto 4 by default;
# main + side + leaf + RCCL's stream + torch's own make more than four)
OUT=gpurun_out/r03z2; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
wall() { grep -o "wall [0-9.]* ms/step" $1 | tail -1; }
for rep in 1 2; do
  timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "plain, default queues: $(wall $OUT/a_$rep.err)"
  GPU_MAX_HW_QUEUES=8 timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "plain, 8 queues: $(wall $OUT/b_$rep.err)"
  timeout 90 $B --force-collectives > $OUT/c_$rep.json 2> $OUT/c_$rep.err; echo "RCCL forced, default queues: $(wall $OUT/c_$rep.err)"
  GPU_MAX_HW_QUEUES=8 timeout 90 $B --force-collectives > $OUT/d_$rep.json 2> $OUT/d_$rep.err; echo "RCCL forced, 8 queues: $(wall $OUT/d_$rep.err)"
done
