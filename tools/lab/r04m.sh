#!/bin/bash
O=gpurun_out/r04m; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for v in 1 0; do
  TPGSR_XBF_HALO3=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2_h3$v.json 2> $O/bench_c3_x2_h3$v.err
  python - <<PY
import json
d=json.load(open("$O/bench_c3_x2_h3$v.json")); r=d["roofline"]
print("HALO3=$v", d["ms_per_step"], "ms/step; family", r["ms_per_step_replayed"], "ms frac", r["frac"], "fwd+dgrad", r["by_kind"]["fwd+dgrad"]["ms"])
for x in r["per_shape"][:10]: print("   ", x)
PY
done
timeout 900 python -m pytest tests/test_tsrn_gpu.py tests/test_crnn_gpu.py tests/test_policy_x2_gpu.py tests/test_schedule_gpu.py -x -q -m gpu 2>&1 | tail -3
