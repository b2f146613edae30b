#!/bin/bash
# how long does the HOST need to submit one C3 step when nothing back-pressures it (few steps), vs the wall time per step
OUT=gpurun_out/host_rate; mkdir -p $OUT
for s in 4 10 30 100; do
  timeout 200 python bench.py --steps $s --warmup 15 --no-cpu-baseline --no-roofline > $OUT/s$s.json 2> $OUT/s$s.err
  echo "steps $s: $(grep 'host submission' $OUT/s$s.err)"
done
