#!/bin/bash
# config 4 on other random tensors than the benched one: ms per step and reads of X per seed
for sd in 1234 1 2 3 4 5 6 7; do
  timeout -k 10 120 python bench.py --seed $sd --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seed', $sd, 'ms_per_step %.2f' % d['ms_per_step'], 'reads', d['roofline']['step']['reads_of_X'])"
done
