#!/bin/bash
# round 3, session A: the one-launch panel kernel -- parity tests, then A/B of the step time
out=gpurun_out/r3a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
B="--no-cpu-baseline --no-other-configs"
timeout -k 10 200 python bench.py --steps 10 --warmup 2 $B > $out/bench_multi.json 2> $out/bench_multi.err; rc=$?
echo "bench multi rc=$rc"; cat $out/bench_multi.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
ok $rc || exit 1
ASB_PANEL_MULTI=0 timeout -k 10 200 python bench.py --steps 10 --warmup 2 $B > $out/bench_old.json 2> $out/bench_old.err; rc=$?
echo "bench old rc=$rc"; cat $out/bench_old.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
ok $rc || exit 1
ASB_DEBUG_PANELS=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 $B > $out/bench_dbg.json 2> $out/bench_dbg.err; rc=$?
grep "multi step\|tile" $out/bench_dbg.err | tail -40
ok $rc || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -m gpu -x > $out/tests.log 2>&1; rc=$?
tail -n 15 $out/tests.log
