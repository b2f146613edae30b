#!/bin/bash
# 1x1 convolutions through the halo kernel: parity, per-shape timing, C3 step
OUT=gpurun_out/r02z; mkdir -p $OUT
TPGSR_XBF_DEBUG=1 timeout 150 python -m pytest tests/test_conv_xbf_gpu.py -m gpu -q -x -p no:cacheprovider -s -k "1x1" > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -c "halo conv" $OUT/tests.log; tail -3 $OUT/tests.log
timeout 120 python tools/lab/halo_1x1.py > $OUT/halo_1x1.md 2>&1; cat $OUT/halo_1x1.md
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
timeout 200 $B > $OUT/c3_default_$rep.json 2> $OUT/err.log; echo "c3 default: $(python -c "import json;d=json.load(open('$OUT/c3_default_$rep.json'));print(d['ms_per_step'])")"
TPGSR_XBF_HALO_MINTAPS=1 timeout 200 $B > $OUT/c3_halo1x1_$rep.json 2>> $OUT/err.log; echo "c3 halo 1x1: $(python -c "import json;d=json.load(open('$OUT/c3_halo1x1_$rep.json'));print(d['ms_per_step'])")"
done
