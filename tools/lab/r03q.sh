#!/bin/bash
# round 3, call q: BatchNorm-backward sums in the epilogue of the producing data-gradient convolution: parity, A/B, the driver's bench command
OUT=gpurun_out/r03q; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_bnb_fuse_gpu.py -m gpu -q -x -p no:cacheprovider -s > $OUT/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $OUT/tests1.log
timeout 400 python -m pytest tests/test_conv_xbf_gpu.py tests/test_conv_panel_gpu.py tests/test_tsrn_gpu.py tests/test_crnn_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 $OUT/tests2.log
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  TPGSR_BNB_FUSE=0 timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "x2, BN-backward sums as their own launch: $(ms $OUT/a_$rep.json)"
  timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "x2, fused into the producing convolution: $(ms $OUT/b_$rep.json)"
done
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_driver_cmd.json
