#!/bin/bash
O=gpurun_out/r04s; mkdir -p $O
python -c "from tpgsr_amd import build as b; assert open(b.LIB+\".stamp\").read()==b._digest(), \"STALE LIBRARY\"" || exit 1
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gru_gate_math_gpu.py tests/test_opt_recorded_gpu.py tests/test_opt_student_gpu.py tests/test_next_models_gpu.py -q -m gpu -x 2>&1 | tail -15
timeout 400 python bench.py --steps 30 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none --tpg opt > $O/bench_c3_opt.json 2> $O/bench_c3_opt.err; tail -3 $O/bench_c3_opt.err
python - <<PY
import json
d=json.load(open("$O/bench_c3_opt.json")); r=d["roofline"]
print("OPT tpg:", d["ms_per_step"], "ms/step", d["value"], "img/s; launches", d["config"]["kernel_launches_per_step"], "family", r["ms_per_step_replayed"], "ms frac", r["frac"])
PY
TPGSR_OPT_RECORD=0 timeout 400 python bench.py --steps 30 --warmup 10 --no-traffic --no-cpu-baseline --no-roofline --alt-prec none --tpg opt > $O/bench_c3_opt_eager.json 2> $O/bench_c3_opt_eager.err
python - <<PY
import json
d=json.load(open("$O/bench_c3_opt_eager.json"))
print("OPT tpg, operator by operator:", d["ms_per_step"], "ms/step", d["value"], "img/s")
PY
