#!/bin/bash
# round 3, session B: k_panel_multi everywhere (DPP reductions, tighter polls) -- timeline, bench, full GPU tests
out=gpurun_out/${1:-r3b}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
B="--no-cpu-baseline --no-other-configs"
timeout -k 10 200 python bench.py --steps 10 --warmup 2 $B > $out/bench.json 2> $out/bench.err; rc=$?
echo "bench rc=$rc"; python -c "import sys,json; d=json.loads(open('$out/bench.json').read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
ok $rc || exit 1
ASB_DEBUG_PANELS=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 $B > $out/bench_dbg.json 2> $out/bench_dbg.err; rc=$?
grep "  step" $out/bench_dbg.err | tail -14
ok $rc || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 bench.py --steps 10 --warmup 2 $B > $out/prof_bench.json 2> $out/prof.err; rc=$?
echo "rocprof rc=$rc"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && head -25 $f | cut -c1-150
ok $rc || exit 1
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $out/tests.log 2>&1; rc=$?
tail -n 8 $out/tests.log
