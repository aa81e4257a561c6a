#!/bin/bash
# sweep the candidate-set size of the projection mode (tuning aid)
for m in "$@"; do
  ASB_M_TARGET=$m python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('M=$m', round(d['value']), round(d['ms_per_step'],2), 'panels', d['roofline']['panels_per_step'])"
done
