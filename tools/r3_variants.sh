#!/bin/bash
# A/B of the 4-tile projection kernel variants on one box: ASB_WIDE_VARIANT = $2.. (4 = default)
out=gpurun_out/${1:-r3v}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="--no-cpu-baseline --no-other-configs"
for v in "$@"; do
  ASB_WIDE_VARIANT=$v timeout -k 10 200 python bench.py --steps 10 --warmup 2 $B > $out/bench_v$v.json 2> $out/bench_v$v.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "variant $v timed out: stopping"; exit 1; }
  python -c "import sys,json; d=json.loads(open('$out/bench_v$v.json').read()); print('variant $v:', round(d['ms_per_step'],3), 'ms/step, launch', round(d['roofline']['avg_launch_ms'],4), 'ms, reads', d['roofline']['step']['reads_of_X'])" || tail -3 $out/bench_v$v.err
done
