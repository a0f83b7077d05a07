#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: python tools/stats_summary.py <csv> [steps] [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(f"total GPU {tot/1e6:.2f} ms, {calls} launches; per step: {tot/1e6/steps:.3f} ms, {calls/steps:.0f} launches")
for r in rows[:top]:
    print(f"{int(r['Calls'])/steps:7.1f}/step {int(r['TotalDurationNs'])/1e3/steps:9.1f} us/step  avg {float(r['AverageNs'])/1e3:8.2f} us  {float(r['Percentage']):5.2f}%  {r['Name'][:95]}")
