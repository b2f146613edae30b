#!/bin/bash
# round 4, call a: schedule harness, x2 gates, the LDS-atomic repro, the per-op time line
set -x
O=gpurun_out/r04a; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
./tools/lab/tail_atomic_repro.bin 200 > $O/tail_atomic_repro.md 2>&1
timeout 900 python -m pytest tests/test_schedule_gpu.py tests/test_policy_x2_gates_gpu.py -x -q -s -m gpu > $O/pytest_new.txt 2>&1
tail -5 $O/pytest_new.txt
timeout 600 python -m pytest tests/test_lstm_seq_gpu.py tests/test_crnn_gpu.py tests/test_rccl_world1_gpu.py -x -q -m gpu > $O/pytest_lstm.txt 2>&1
tail -3 $O/pytest_lstm.txt
timeout 300 python tools/plan_gaps.py --config c3 --prec x2 --out $O/plan_gaps_c3_x2.md > $O/plan_gaps.log 2>&1
tail -30 $O/plan_gaps.log
timeout 300 python bench.py --steps 40 --warmup 10 --no-traffic --no-cpu-baseline --alt-prec none > $O/bench_c3_x2.json 2> $O/bench_c3_x2.err
cat $O/bench_c3_x2.json | head -c 600
