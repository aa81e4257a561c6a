#!/usr/bin/env python3
"""Where the GPU sits idle between kernels: from a rocprofv3 --kernel-trace SQLite result (*_results.db), the dispatch timeline of
the busiest window, idle time grouped by the pair (kernel before the gap, kernel after it).

    python tools/rocpd_gaps.py gpurun_out/<tag>/prof/stats_results.db [first_kernel_substring] [min_gap_us] [end_kernel_substring]
The window starts at the LAST dispatch of `first_kernel_substring` minus nothing (default: the whole trace)."""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
names = dict(con.execute("select id, kernel_name from kernel_symbols")) if True else {}
try:
    rows = list(con.execute("select kernel_id, start, end from rocpd_kernel_dispatch order by start"))
except sqlite3.OperationalError:
    rows = []
ks = {}
for kid, name in con.execute("select id, display_name from rocpd_info_kernel_symbol"):
    ks[kid] = name.split("(")[0][:44]
disp = [(ks.get(k, str(k)), s, e) for k, s, e in rows]
sub = sys.argv[2] if len(sys.argv) > 2 else None
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
end_sub = sys.argv[4] if len(sys.argv) > 4 else None
if sub:
    idx = [i for i, d in enumerate(disp) if sub in d[0]]
    if idx:
        # with an END substring: the LAST window [last `sub` .. first `end_sub` behind it] (one timed step of a bench run)
        start = idx[-1] if end_sub else idx[0]
        disp = disp[start:]
        if end_sub:
            e = [i for i, d in enumerate(disp) if end_sub in d[0]]
            if e:
                disp = disp[:e[0] + 1]
t0, t1 = disp[0][1], max(d[2] for d in disp)
busy = 0
gaps = defaultdict(lambda: [0, 0.0])
cur_end = disp[0][2]
busy += disp[0][2] - disp[0][1]
prev = disp[0][0]
for name, s, e in disp[1:]:
    if s > cur_end:
        g = (s - cur_end) / 1e3
        if g >= min_gap:
            gaps[(prev, name)][0] += 1
            gaps[(prev, name)][1] += g
        busy += e - s
    else:
        busy += max(0, e - cur_end)
    if e > cur_end:
        cur_end = e
        prev = name
print("window %.2f ms, kernels busy %.2f ms, idle %.2f ms" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
for (a, b), (n, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%8.1f us in %4d gaps (avg %6.1f)  %-44s -> %s" % (tot, n, tot / n, a, b))
