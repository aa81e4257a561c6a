#!/bin/bash
out=gpurun_out/${1:-r4c5}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ASB_DEBUG_DEIM=1 timeout -k 10 400 python tools/time_c5.py 3 2>&1 | grep -v "^Function\|amdgpu.ids" | tail -8
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof5 -o stats -- python3 tools/time_c5.py 1 > $out/c5.log 2> $out/prof5.err
python tools/rocpd_stats.py $out/prof5/stats_results.db > $out/c5_kernel_stats.csv; head -22 $out/c5_kernel_stats.csv | cut -c1-150
