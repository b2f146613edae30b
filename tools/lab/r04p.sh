#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
python -c "from tpgsr_amd import build as b; assert open(b.LIB+\".stamp\").read()==b._digest(), \"STALE LIBRARY\"" || exit 1
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_rccl_world1_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --eval --steps 40 --warmup 10 > $O/bench_eval.json 2> $O/bench_eval.err; tail -3 $O/bench_eval.err
python - <<PY
import json
d=json.load(open("$O/bench_eval.json")); r=d["roofline"]
print("EVAL", d["value"], "img/s", d["ms_per_step"], "ms/batch; sr only", d["super_resolve_only"], "family", r["ms_per_step_replayed"], "frac", r["frac"], "cpu", d.get("cpu_baseline"))
for x in r["per_shape"][:8]: print("   ", x)
PY
for mw in 16384 4096; do
TPGSR_XBF_WGRAD_HALO_MINWORK=$mw timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_mw$mw.json 2> $O/bench_mw$mw.err
python - <<PY
import json
d=json.load(open("$O/bench_mw$mw.json")); r=d["roofline"]
print("MINWORK $mw:", d["ms_per_step"], "ms/step; family", r["ms_per_step_replayed"], "ms frac", r["frac"], {k:v["ms"] for k,v in r["by_kind"].items()}, d.get("eval"))
for x in r["per_shape"][:16]:
    if x["kind"].startswith("wgrad"): print("   ", x)
PY
done
