#!/bin/bash
set -x
O=gpurun_out/r04f; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_conv_halo3_gpu.py -x -q -m gpu > $O/pytest_a.txt 2>&1; tail -4 $O/pytest_a.txt
timeout 120 python tools/lab/halo3_trace.py 48 16 64 64 64 3 3 1 2 > $O/trace_trunk.txt 2>&1
head -16 $O/trace_trunk.txt
for sh in "48 16 64 64 256" "48 8 25 256 256" "48 4 26 512 512" "48 16 50 64 128" "48 8 25 128 256" "48 4 26 256 512"; do
  timeout 120 python tools/lab/halo3_trace.py $sh 3 3 1 2 2>&1 | grep "====" | sed "s/^/$sh: /" | tee -a $O/trace_others.txt
done
for v in 1 0; do
  TPGSR_XBF_HALO3=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2_h3$v.json 2> $O/bench_c3_x2_h3$v.err
  python - <<PY
import json
d=json.load(open("$O/bench_c3_x2_h3$v.json")); r=d["roofline"]
print("HALO3=$v", d["ms_per_step"], "ms/step; family", r["ms_per_step_replayed"], "ms frac", r["frac"])
for x in r["per_shape"][:8]: print("   ", x)
PY
done
