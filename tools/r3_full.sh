#!/bin/bash
# round 3 full session: GPU tests, default bench, kernel statistics (config 4 and config 5), PMC traffic passes
tag=${1:-r3full}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $out/tests.log 2>&1; rc=$?
tail -n 6 $out/tests.log
ok $rc || { echo "tests timed out: stopping"; exit 1; }
timeout -k 10 700 python bench.py > $out/bench.json 2> $out/bench.err; rc=$?
echo "bench rc=$rc"; python -c "import json; d=json.loads(open('$out/bench.json').read()); print(d['ms_per_step'], d['roofline']['frac'], d['end_to_end']['total_ms'], {k:(v.get('ms'),v.get('prepare_ms')) for k,v in d['other_configs'].items()})"
ok $rc || exit 1
B="--no-cpu-baseline --no-other-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 bench.py --steps 10 --warmup 2 $B > $out/prof_bench.json 2> $out/prof.err; rc=$?
echo "rocprof rc=$rc"; python tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.csv; head -12 $out/kernel_stats.csv | cut -c1-140
ok $rc || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof5 -o stats -- python3 tools/profile_c5.py > $out/c5.json 2> $out/prof5.err; rc=$?
python tools/rocpd_stats.py $out/prof5/stats_results.db > $out/c5_kernel_stats.csv; cat $out/c5.json | cut -c1-200
ok $rc || exit 1
bash tools/pmc_session.sh $tag
python tools/summarise_pmc.py $out/pmc_fetch $out/pmc_write r03tmp && mv profiles/r03tmp_pmc_traffic.json $out/pmc_traffic.json
