#!/bin/bash
# round 4, call b: LSTM poison rework, fused GruBlock weight gradients (parity + A/B), the corrected per-op time line
set -x
O=gpurun_out/r04b; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_lstm_seq_gpu.py tests/test_gru_wgrad_gpu.py -x -q -s -m gpu > $O/pytest_a.txt 2>&1
tail -4 $O/pytest_a.txt
timeout 900 python -m pytest tests/test_tsrn_gpu.py tests/test_blocks_gpu.py tests/test_crnn_gpu.py tests/test_schedule_gpu.py tests/test_policy_x2_gpu.py -x -q -m gpu > $O/pytest_b.txt 2>&1
tail -4 $O/pytest_b.txt
for v in 1 0; do
  TPGSR_GRU_WGRAD=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2_gw$v.json 2> $O/bench_c3_x2_gw$v.err
  python - <<PY
import json
d=json.load(open("$O/bench_c3_x2_gw$v.json")); r=d["roofline"]
print("GRU_WGRAD=$v", d["ms_per_step"], "ms/step; family", r["ms_per_step_replayed"], "ms frac", r["frac"], "launches", r["launches_per_step"])
for x in r["per_shape"][:12]: print("   ", x)
PY
done
TPGSR_GRU_WGRAD_Z=256 timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2_gwz256.json 2> $O/bench_c3_x2_gwz256.err
python -c "
import json; d=json.load(open('$O/bench_c3_x2_gwz256.json')); print('Z=256', d['ms_per_step'], d['roofline']['ms_per_step_replayed'])"
timeout 300 python tools/plan_gaps.py --config c3 --prec x2 --out $O/plan_gaps_c3_x2.md > $O/plan_gaps.log 2>&1
head -16 $O/plan_gaps.log
