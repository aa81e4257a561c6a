#!/bin/bash
# One GPU-box session: GPU tests, the default bench, kernel statistics of the bench.  Steps after a timed-out / killed
# step are skipped (a hung GPU step must not be followed by another).  Usage: tools/gpu_session.sh <tag> [pytest args]
tag=${1:-sess}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 900 python -m pytest tests -q -m gpu "$@" > $out/tests.log 2>&1; rc=$?
tail -n 15 $out/tests.log
ok $rc || { echo "tests timed out: stopping"; exit 1; }
timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err; rc=$?
echo "bench rc=$rc"; cat $out/bench.json
ok $rc || { echo "bench timed out: stopping"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $out/prof_bench.json 2> $out/prof.err; rc=$?
echo "rocprof rc=$rc"; cat $out/prof_bench.json
find $out/prof -name "*kernel_stats.csv" | head -2
