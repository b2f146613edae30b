#!/bin/bash
set -x
O=gpurun_out/r04d; mkdir -p $O
timeout 300 python -m pytest tests/test_pack_once_gpu.py -x -q -m gpu > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 120 python tools/lab/halo3_trace.py 48 16 64 64 64 3 3 1 2 > $O/trace_trunk.txt 2>&1
cat $O/trace_trunk.txt
timeout 120 python tools/lab/halo3_trace.py 48 16 64 64 256 3 3 1 2 > $O/trace_up.txt 2>&1
head -30 $O/trace_up.txt
