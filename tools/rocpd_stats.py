#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes its --kernel-trace --stats result as an SQLite file (*_results.db): dumps the per-kernel
summary (calls, total / average duration in microseconds, share) as CSV -- what is committed under profiles/.

    python tools/rocpd_stats.py gpurun_out/<tag>/prof/stats_results.db > profiles/<name>_kernel_stats.csv
"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    w.writerow([name, calls, "%.1f" % total, "%.2f" % avg, "%.2f" % pct])
