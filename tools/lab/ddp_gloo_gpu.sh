#!/bin/bash
# two ranks of the TPGSR step on ONE GPU over gloo, each rank's output in its own log (debugging aid for tests/test_ddp_gpu.py)
OUT=gpurun_out/ddp; mkdir -p $OUT
for R in 0 1; do
  timeout 300 python - $R > $OUT/rank$R.log 2>&1 <<'PY' &
import sys, os, queue
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import faulthandler; faulthandler.enable()
import test_ddp_gpu as T
class Q:
    def put(self, x): print("RESULT", x[0], type(x[1]), (x[1][:2000] if isinstance(x[1], str) else x[1].shape))
T._worker(int(sys.argv[1]), 2, 32123, Q(), True)
print("done")
PY
done
wait
tail -25 $OUT/rank0.log; echo ----; tail -12 $OUT/rank1.log
