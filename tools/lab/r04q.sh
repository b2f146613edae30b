#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O
python -c "from tpgsr_amd import build as b; assert open(b.LIB+\".stamp\").read()==b._digest(), \"STALE LIBRARY\"" || exit 1
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gru_gate_math_gpu.py tests/test_kernels_gpu.py tests/test_conv_halo3_gpu.py tests/test_bnb_fuse_gpu.py tests/test_gru_wgrad_gpu.py tests/test_blocks_gpu.py tests/test_tsrn_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2.json 2> $O/bench_c3_x2.err
python - <<PY
import json
d=json.load(open("$O/bench_c3_x2.json")); r=d["roofline"]
print(d["ms_per_step"], "ms/step; family", r["ms_per_step_replayed"], "ms frac", r["frac"], {k:v["ms"] for k,v in r["by_kind"].items()})
for x in r["per_shape"][:12]: print("   ", x)
PY
timeout 300 python tools/plan_gaps.py --out $O/plan_gaps_c3_x2.md > $O/plan_gaps.log 2>&1; head -12 $O/plan_gaps_c3_x2.md
