#!/bin/bash
# round 3, call x: the driver's multi-rank launch line on ONE GPU (TPGSR_BENCH_SHARED_GPU=1: ranks share cuda:0, gloo) -- a self-test of the N > 1 code
# path of bench.py after this round's changes, not a measurement
OUT=gpurun_out/r03x; mkdir -p $OUT
export TMPDIR=/tmp
TPGSR_BENCH_SHARED_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_2ranks_shared.json 2> $OUT/bench_2ranks_shared.err; echo "2 ranks rc=$?"
cut -c1-700 $OUT/bench_2ranks_shared.json; grep -v amdgpu.ids $OUT/bench_2ranks_shared.err | tail -8
