#!/bin/bash
# config 3 A/B: the SPLOCS loop with the objective trace on the device (default) / read back every iteration, then kernel statistics
out=gpurun_out/${1:-r4c3}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for rep in 1 2; do
  ASB_SPLOCS_DEFER=1 timeout -k 10 300 python tools/time_c3.py 2>/dev/null | tail -1
  ASB_SPLOCS_DEFER=0 timeout -k 10 300 python tools/time_c3.py 2>/dev/null | tail -1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof3 -o stats -- python3 tools/time_c3.py > $out/c3.log 2> $out/prof3.err; rc=$?
python tools/rocpd_stats.py $out/prof3/stats_results.db > $out/c3_kernel_stats.csv; head -25 $out/c3_kernel_stats.csv | cut -c1-150
