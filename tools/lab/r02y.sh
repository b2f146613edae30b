#!/bin/bash
# experiment batch: pack-aside, 1x1 through the halo kernel, BN backward reduce
OUT=gpurun_out/r02y; mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv_xbf_gpu.py tests/test_kernels_gpu.py tests/test_crnn_gpu.py tests/test_tsrn_gpu.py -m gpu -q -x -p no:cacheprovider -k "1x1 or bn or reduce or batch or norm or step or traject or train or cascade" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 200 python tools/lab/halo_1x1.py > $OUT/halo_1x1.md 2>&1; cat $OUT/halo_1x1.md
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
timeout 300 $B > $OUT/c3_default_$rep.json 2> $OUT/err.log; echo "c3 default: $(python -c "import json;d=json.load(open('$OUT/c3_default_$rep.json'));print(d['ms_per_step'])")"
TPGSR_PACK_ASIDE=0 timeout 300 $B > $OUT/c3_nopackaside_$rep.json 2>> $OUT/err.log; echo "c3 pack in plan: $(python -c "import json;d=json.load(open('$OUT/c3_nopackaside_$rep.json'));print(d['ms_per_step'])")"
TPGSR_XBF_HALO_MINTAPS=1 timeout 300 $B > $OUT/c3_halo1x1_$rep.json 2>> $OUT/err.log; echo "c3 halo 1x1: $(python -c "import json;d=json.load(open('$OUT/c3_halo1x1_$rep.json'));print(d['ms_per_step'])")"
done
timeout 300 $B --config c5 --steps 30 --warmup 8 > $OUT/c5_default.json 2>> $OUT/err.log; echo "c5 default: $(python -c "import json;d=json.load(open('$OUT/c5_default.json'));print(d['ms_per_step'])")"
TPGSR_PACK_ASIDE=0 timeout 300 $B --config c5 --steps 30 --warmup 8 > $OUT/c5_nopackaside.json 2>> $OUT/err.log; echo "c5 pack in plan: $(python -c "import json;d=json.load(open('$OUT/c5_nopackaside.json'));print(d['ms_per_step'])")"
tail -5 $OUT/err.log
