#!/bin/bash
OUT=gpurun_out/r02m; mkdir -p $OUT
timeout 240 python -m pytest tests/test_moran_gpu.py tests/test_aster_gpu.py -m gpu -q -x -s -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -v "^$" $OUT/tests.log | tail -25
