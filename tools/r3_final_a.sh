#!/bin/bash
# final session, part A: all GPU tests, then the default bench line (what the driver runs)
tag=${1:-r3fa}
out=gpurun_out/$tag
mkdir -p $out
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 700 python -m pytest tests -q -m gpu -x > $out/tests.log 2>&1; rc=$?
tail -n 4 $out/tests.log
ok $rc || { echo "tests timed out: stopping"; exit 1; }
timeout -k 10 420 python bench.py > $out/bench.json 2> $out/bench.err; rc=$?
echo "bench rc=$rc"; python -c "import json; d=json.loads(open('$out/bench.json').read()); print(d['ms_per_step'], d['roofline']['frac'], d['end_to_end']['total_ms'], {k:(round(v.get('ms',0),2),round(v.get('prepare_ms',0),1)) for k,v in d['other_configs'].items()})"
