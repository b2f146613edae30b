#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
for d in 0 2 4 6; do
  echo "=== TPGSR_H3_DEBUG=$d"
  TPGSR_H3_DEBUG=$d timeout 120 python tools/lab/halo3_trace.py 48 16 64 64 64 3 3 1 2 2>&1 | head -8 | tee -a $O/dbg.txt
done
