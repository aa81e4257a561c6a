#!/bin/bash
# the multi-rank protocol forced on one rank beside the fused driver: step time, kernel statistics, idle gaps
out=gpurun_out/${1:-r4forced}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="--steps 20 --warmup 3 --no-cpu-baseline --no-other-configs"
for f in 0 1; do
  ASB_FORCE_COLLECTIVES=$f timeout -k 10 300 python bench.py $B > $out/bench_f$f.json 2> $out/bench_f$f.err
  python -c "import json; d=json.loads(open('$out/bench_f$f.json').read()); print('forced=$f', round(d['ms_per_step'],3), 'ms per step')"
done
export ASB_FORCE_COLLECTIVES=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_f1 -o stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $out/prof_f1.json 2> $out/prof_f1.err
python tools/rocpd_stats.py $out/prof_f1/stats_results.db > $out/kernel_stats_forced.csv
python tools/rocpd_gaps.py $out/prof_f1/stats_results.db k_begin_reset 4 | head -30
