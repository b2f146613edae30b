#!/bin/bash
OUT=gpurun_out/r02i; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_a-z]+" | sort -u > $OUT/counters.txt; wc -l $OUT/counters.txt
for M in x3 bf16; do
  echo "=== conv5 512->512 $M pipe0"; TPGSR_XBF_PIPE=0 bash tools/lab/pmc_conv.sh $OUT/c5_${M}_p0 "48 4 26 512 512 3 3 1" $M
  echo "=== conv5 512->512 $M pipe64"; TPGSR_XBF_PIPE=64 bash tools/lab/pmc_conv.sh $OUT/c5_${M}_p64 "48 4 26 512 512 3 3 1" $M
done
echo "=== 3x3 64->64 x3 pipe0"; TPGSR_XBF_PIPE=0 bash tools/lab/pmc_conv.sh $OUT/t_x3_p0 "48 16 64 64 64 3 3 1" x3
echo "=== 3x3 64->64 x3 pipe64"; TPGSR_XBF_PIPE=64 bash tools/lab/pmc_conv.sh $OUT/t_x3_p64 "48 16 64 64 64 3 3 1" x3
find $OUT -name "*.csv" -size +2M -delete
