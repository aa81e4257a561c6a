#!/bin/bash
# sketch predictor session: its tests, the low-rank tests of round 2, the c4_lowrank leg with the per-panel log
out=gpurun_out/${1:-sk}
mkdir -p $out
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 600 python -m pytest tests/test_gpu_sketch.py -q -m gpu -x -s > $out/tests.log 2>&1; rc=$?
tail -n 15 $out/tests.log
ok $rc || exit 1
[ $rc -eq 0 ] || exit 1
ASB_DEBUG_PANELS=1 timeout -k 10 300 python tools/lowrank_probe.py > $out/lowrank.log 2> $out/lowrank.err; rc=$?
grep -v "  step" $out/lowrank.err | grep "asb\]" | cut -c1-150 | tail -40; tail -n 2 $out/lowrank.log
ok $rc || exit 1
timeout -k 10 200 python tools/lowrank_probe.py > $out/lowrank2.log 2>&1; tail -n 1 $out/lowrank2.log
