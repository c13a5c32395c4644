#!/usr/bin/env python
"""Aggregate a rocprofv3 counter_collection.csv per kernel name: launches, sum and mean of one counter.
usage: pmc_aggregate.py <counter_collection.csv> <COUNTER>  -> CSV on stdout (sorted by total, descending)"""
import csv
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
tot, n = defaultdict(float), defaultdict(int)
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != counter:
            continue
        k = row["Kernel_Name"]
        tot[k] += float(row["Counter_Value"])
        n[k] += 1
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Launches", counter + "_total", counter + "_per_launch"])
for k in sorted(tot, key=tot.get, reverse=True):
    w.writerow([k, n[k], "%.1f" % tot[k], "%.3f" % (tot[k] / n[k])])
