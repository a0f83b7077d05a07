#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.
usage: pmc_summary.py <counter_collection.csv> [name-substring ...]"""
import csv
import sys
from collections import defaultdict

rows = csv.DictReader(open(sys.argv[1]))
filters = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
    if filters and not any(f in name for f in filters):
        continue
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, counters in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    # med = the typical dispatch of a training step (the mean mixes in the first step's launches over the whole arena)
    parts = [f"{c}: n={len(v)} mean={sum(v)/len(v):.1f} max={max(v):.1f} med={sorted(v)[len(v) // 2]:.1f}" for c, v in counters.items()]
    print(name[:110], "|", " ; ".join(parts))
