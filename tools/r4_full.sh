#!/bin/bash
# round 4 full session: GPU tests, default bench (all legs + CPU baselines), kernel statistics (config 4, forced collectives, config 3,
# config 5), PMC traffic passes, MFMA-busy pass.   usage: tools/r4_full.sh TAG [skip_tests]
tag=${1:-r4full}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
if [ -z "$2" ]; then
  timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $out/tests.log 2>&1; rc=$?
  tail -n 6 $out/tests.log
  ok $rc || { echo "tests timed out: stopping"; exit 1; }
fi
timeout -k 10 1000 python bench.py > $out/bench.json 2> $out/bench.err; rc=$?
echo "bench rc=$rc"; python - <<PY
import json
d=json.loads(open('$out/bench.json').read())
print('headline', round(d['ms_per_step'],3), 'ms; roofline frac', round(d['roofline']['frac'],3), 'launch', round(d['roofline']['avg_launch_ms'],4), 'e2e', round(d['end_to_end'].get('total_ms',0),2), 'cpu', d.get('cpu_baseline',{}).get('value'))
for k,v in d.get('other_configs',{}).items():
    print(k, {q:(round(v[q],2) if isinstance(v[q],float) else v[q]) for q in ('ms','cold_ms','prepare_ms','reads_of_X','residual_mode_ms','pod_ms','post_process_ms','deim_ms','error') if q in v})
PY
ok $rc || exit 1
B="--no-cpu-baseline --no-other-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 bench.py --steps 10 --warmup 2 $B > $out/prof_bench.json 2> $out/prof.err; rc=$?
echo "rocprof rc=$rc"; python tools/rocpd_stats.py $out/prof/stats_results.db > $out/kernel_stats.csv; head -10 $out/kernel_stats.csv | cut -c1-140
python tools/rocpd_gaps.py $out/prof/stats_results.db k_begin_reset 4 k_publish_results | head -12
ok $rc || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof5 -o stats -- python3 tools/time_c5.py 1 > $out/c5.log 2> $out/prof5.err; rc=$?
python tools/rocpd_stats.py $out/prof5/stats_results.db > $out/kernel_stats_config5.csv; tail -2 $out/c5.log | cut -c1-300
ok $rc || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof3 -o stats -- python3 tools/time_c3.py > $out/c3.log 2> $out/prof3.err; rc=$?
python tools/rocpd_stats.py $out/prof3/stats_results.db > $out/kernel_stats_config3.csv; tail -1 $out/c3.log
ok $rc || exit 1
bash tools/pmc_session.sh $tag
python tools/summarise_pmc.py $out/pmc_fetch $out/pmc_write r04tmp && mv profiles/r04tmp_pmc_traffic.json $out/pmc_traffic.json
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 1 --warmup 0 $B > $out/pmc_mfma.json 2> $out/pmc_mfma.err; rc=$?
echo "pmc mfma rc=$rc"; python tools/summarise_pmc_mfma.py $out/pmc_mfma r04tmp && mv profiles/r04tmp_pmc_mfma.json $out/pmc_mfma_summary.json
