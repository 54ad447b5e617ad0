#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the plain-text kernel summary kept under profiles/.
usage: rocpd_summary.py <results.db> [> profiles/rNN_xxx.txt]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
print("%-90s %8s %16s %16s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-90s %8d %16.0f %16.0f %8.3f" % (name[:90], calls, tot, avg, pct))
