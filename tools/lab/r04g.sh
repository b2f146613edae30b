#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
for v in 1 0; do
  echo "=== HALO3=$v"
  TPGSR_XBF_HALO3=$v bash tools/lab/pmc_conv.sh $O/h3_$v "48 16 64 64 64 3 3 1" x2 2>&1 | grep -v "^$" | tee $O/pmc_h3_$v.txt
done
