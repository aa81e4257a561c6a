#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   (other env vars are inherited) -- prints value / ms per step / avg launch ms
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$var=$v', round(d['value']), round(d['ms_per_step'],2), 'avg_launch_ms', round(r['avg_launch_ms'],4), 'launches', r['launches'], 'panels', r['panels_per_step'])"
done
