#!/bin/bash
# quick A/B: step time + panel-kernel timeline (+ optional pytest selection as further args)
out=gpurun_out/${1:-r3q}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
B="--no-cpu-baseline --no-other-configs"
timeout -k 10 200 python bench.py --steps 10 --warmup 2 $B > $out/bench.json 2> $out/bench.err; rc=$?
echo "bench rc=$rc"; python -c "import sys,json; d=json.loads(open('$out/bench.json').read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
ok $rc || exit 1
ASB_DEBUG_PANELS=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 $B > $out/bench_dbg.json 2> $out/bench_dbg.err; rc=$?
grep "  step" $out/bench_dbg.err | tail -8
ok $rc || exit 1
if [ $# -gt 0 ]; then
  timeout -k 10 900 python -m pytest "$@" -q -m gpu -x > $out/tests.log 2>&1; rc=$?
  tail -n 6 $out/tests.log
fi
