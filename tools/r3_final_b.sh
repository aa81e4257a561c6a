#!/bin/bash
# final session, part B: kernel statistics of the benched build (config 4, the low-rank leg, config 5), PMC traffic passes
tag=${1:-r3fb}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
B="--no-cpu-baseline --no-other-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 bench.py --steps 10 --warmup 2 $B > $out/prof_bench.json 2> $out/prof.err; rc=$?
echo "rocprof rc=$rc"; python tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.csv; sed -n 1,8p $out/kernel_stats.csv | cut -c1-140
ok $rc || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/proflr -o stats -- python3 tools/lowrank_probe.py > $out/lowrank.log 2> $out/proflr.err; rc=$?
python tools/rocpd_stats.py $out/proflr/stats_results.db > $out/kernel_stats_lowrank.csv; tail -n 1 $out/lowrank.log
ok $rc || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof5 -o stats -- python3 tools/profile_c5.py > $out/c5.json 2> $out/prof5.err; rc=$?
python tools/rocpd_stats.py $out/prof5/stats_results.db > $out/kernel_stats_config5.csv; cut -c1-200 $out/c5.json
ok $rc || exit 1
bash tools/pmc_session.sh $tag
python tools/summarise_pmc.py $out/pmc_fetch $out/pmc_write r03tmp && mv profiles/r03tmp_pmc_traffic.json $out/pmc_traffic.json
