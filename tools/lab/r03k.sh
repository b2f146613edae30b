#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout 120 python tools/lab/step_phases.py > $OUT/phases.md 2> $OUT/phases.err; echo "rc=$?"; cat $OUT/phases.md; tail -3 $OUT/phases.err
