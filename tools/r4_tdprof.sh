#!/bin/bash
# per-launch durations of the panel tridiagonalisation (k_td_panel: 32 reflectors per launch) against the trailing size
out=gpurun_out/${1:-r4td}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof -o stats -- python3 tools/time_tridiag.py 4000 > $out/run.log 2> $out/prof.err
tail -1 $out/run.log
python - <<PY
import sqlite3
con=sqlite3.connect('$out/prof/stats_results.db')
ks={k:n.split('(')[0][:40] for k,n in con.execute("select id, display_name from rocpd_info_kernel_symbol")}
rows=[(ks[k],s,e) for k,s,e in con.execute("select kernel_id,start,end from rocpd_kernel_dispatch order by start")]
pan=[(e-s)/1e3 for n,s,e in rows if 'k_td_panel' in n]
r2=[(e-s)/1e3 for n,s,e in rows if 'k_td_rank2k' in n]
per=len(pan)//3
print('panels per call', per)
p=pan[-per:]; q=r2[-per:]
for i in range(0,per,8): print('panel %3d (trailing %4d): %7.1f us = %5.2f us per reflector; rank-2k %6.1f us'%(i,4000-32*i,p[i],p[i]/32,q[i]))
print('sum panels %.1f ms, rank2k %.1f ms'%(sum(p)/1e3,sum(q)/1e3))
PY
