#!/bin/bash
# round 4 session: [pytest selection] -> bench legs -> probes.   usage: tools/r4_session.sh TAG "pytest args" "bench legs" [probe kinds]
tag=${1:-r4s}; sel=${2:-}; legs=${3:-}; kinds=${4:-}
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
if [ -n "$sel" ]; then
  timeout -k 10 1100 python -m pytest $sel -q -m gpu -x > $out/tests.log 2>&1; rc=$?
  tail -n 25 $out/tests.log; ok $rc || { echo "tests timed out: stopping"; exit 1; }
fi
if [ -n "$legs" ]; then
  timeout -k 10 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --legs "$legs" > $out/bench.json 2> $out/bench.err; rc=$?
  echo "bench rc=$rc"; python - <<PY
import json
d=json.loads(open('$out/bench.json').read())
print('headline', round(d['ms_per_step'],3), 'ms, roofline frac', round(d['roofline']['frac'],3), 'launch', round(d['roofline']['avg_launch_ms'],4))
for k,v in d.get('other_configs',{}).items():
    print(k, {q:(round(v[q],2) if isinstance(v[q],float) else v[q]) for q in ('ms','reads_of_X','sketch_replays','residual_switch_at','residual_mode_ms','error','each','ms_each','panel_kernel_fallbacks') if q in v})
PY
  ok $rc || exit 1
fi
if [ -n "$kinds" ]; then
  timeout -k 10 400 python tools/structured_probe.py $kinds > $out/probe.log 2> $out/probe.err; rc=$?
  cat $out/probe.log; ok $rc || exit 1
  ASB_DEBUG_PANELS=1 timeout -k 10 400 python tools/structured_probe.py $kinds > $out/probe_dbg.log 2> $out/probe_dbg.err; rc=$?
  ok $rc || exit 1
fi
