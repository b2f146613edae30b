#!/bin/bash
# final measurement set of the round (one gpurun call)
T=$1
bash tools/gpu_run.sh $T tests_all smoke
bash tools/gpu_run.sh $T bench_c3
bash tools/gpu_run.sh $T prof_c3 plan_gaps
BENCH_ARGS="--no-cpu-baseline --no-traffic" bash tools/gpu_run.sh $T bench_eval
bash tools/gpu_run.sh $T bench_c2 bench_c5
timeout 200 python tools/lab/shape_table.py c3 > gpurun_out/$T/shape_table_c3.json 2> gpurun_out/$T/shape_table_c3.err
ls gpurun_out/$T
