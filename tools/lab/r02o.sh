#!/bin/bash
OUT=gpurun_out/r02o; mkdir -p $OUT
timeout 100 python -m pytest tests/test_moran_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
bash tools/gpu_run.sh r02o prof_c2 > $OUT/prof.log 2>&1; tail -3 $OUT/prof.log
