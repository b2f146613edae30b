#!/bin/bash
# round 3, call ae: the train-step test files and smoke() at the final HEAD (the full suite ran one commit earlier: profiles/r03ac_gpu_suite.txt)
OUT=gpurun_out/r03ae; mkdir -p $OUT
export TMPDIR=/tmp
timeout 170 python -m pytest tests/test_crnn_gpu.py tests/test_policy_x2_gpu.py tests/test_rccl_world1_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
