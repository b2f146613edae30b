#!/bin/bash
# round 4, call c: whole-CU halo kernel (parity, A/B), eval-mode pack-once
set -x
O=gpurun_out/r04c; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_conv_halo3_gpu.py tests/test_pack_once_gpu.py -x -q -s -m gpu > $O/pytest_a.txt 2>&1
tail -6 $O/pytest_a.txt
for v in 1 0; do
  TPGSR_XBF_HALO3=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2_h3$v.json 2> $O/bench_c3_x2_h3$v.err
  python - <<PY
import json
d=json.load(open("$O/bench_c3_x2_h3$v.json")); r=d["roofline"]
print("HALO3=$v", d["ms_per_step"], "ms/step; family", r["ms_per_step_replayed"], "ms frac", r["frac"])
for x in r["per_shape"][:16]: print("   ", x)
PY
done
TPGSR_XBF_HALO3_MIN=256 timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2_h3min256.json 2> $O/bench_c3_x2_h3min256.err
python -c "
import json; d=json.load(open('$O/bench_c3_x2_h3min256.json')); print('HALO3_MIN=256', d['ms_per_step'], d['roofline']['ms_per_step_replayed'])"
TPGSR_PACK_ALWAYS=1 timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none --no-roofline > $O/bench_c3_x2_packalways.json 2> $O/bench_c3_x2_packalways.err
python -c "
import json; d=json.load(open('$O/bench_c3_x2_packalways.json')); print('PACK_ALWAYS', d['ms_per_step'])"
timeout 200 python tools/lab/shape_table.py c3 > $O/shape_table_c3_x2.json 2>/dev/null
