#!/bin/bash
set -x
O=gpurun_out/r04e; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_halo3_gpu.py -x -q -m gpu > $O/pytest_a.txt 2>&1; tail -4 $O/pytest_a.txt
timeout 120 python tools/lab/halo3_trace.py 48 16 64 64 64 3 3 1 2 > $O/trace_trunk.txt 2>&1
head -16 $O/trace_trunk.txt
timeout 120 python tools/lab/halo3_trace.py 48 16 64 64 256 3 3 1 2 > $O/trace_up.txt 2>&1
grep "====" $O/trace_up.txt
timeout 120 python tools/lab/halo3_trace.py 48 8 25 256 256 3 3 1 2 > $O/trace_c3.txt 2>&1
grep "====" $O/trace_c3.txt
timeout 120 python tools/lab/halo3_trace.py 48 4 26 512 512 3 3 1 2 > $O/trace_c5.txt 2>&1
grep "====" $O/trace_c5.txt
timeout 120 python tools/lab/halo3_trace.py 48 16 50 64 128 3 3 1 2 > $O/trace_c1.txt 2>&1
grep "====" $O/trace_c1.txt
