#!/usr/bin/env python3
"""Turns rocprofv3's rocpd sqlite output (ROCm 7.2 default) into the plain-text per-kernel summary we
commit under profiles/.  Usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys

for db in sys.argv[1:]:
    cur = sqlite3.connect(db).cursor()
    print(f"# {db}")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{calls:>6} {total:>12.1f} {avg:>10.2f} {pct:>6.2f}  {name[:150]}")
    print()
