#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
python -c "from tpgsr_amd import build as b; assert open(b.LIB+\".stamp\").read()==b._digest(), \"STALE LIBRARY\"" || exit 1
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv_xbf_gpu.py tests/test_conv_panel_gpu.py tests/test_conv_halo3_gpu.py tests/test_gru_wgrad_gpu.py tests/test_bnb_fuse_gpu.py tests/test_wgrad_reduce_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2.json 2> $O/bench_c3_x2.err
python - <<PY
import json
d=json.load(open("$O/bench_c3_x2.json")); r=d["roofline"]
print(d["ms_per_step"], "ms/step; family", r["ms_per_step_replayed"], "ms frac", r["frac"], {k:v["ms"] for k,v in r["by_kind"].items()})
for x in r["per_shape"][:16]: print("   ", x)
PY
