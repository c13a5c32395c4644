#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE per-kernel aggregates (tools/pmc_aggregate.py) -> profiles/r01_pmc_hbm_traffic.json.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters report KiB-sized units and FETCH_SIZE
counts 64-byte requests as 32 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).
usage: pmc_combine.py <FETCH_SIZE.csv> <WRITE_SIZE.csv> <out.json> [note]"""
import csv
import json
import sys


def read(path, col):
    out = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            out[r["Kernel_Name"]] = (int(r["Launches"]), float(r[col + "_per_launch"]))
    return out


fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
kernels = {}
for name, (n, f) in fetch.items():
    w = write.get(name, (n, 0.0))[1]
    kernels[name] = {"launches": n, "fetch_kib_raw": f, "write_kib_raw": w,
                     "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
json.dump({"note": sys.argv[4] if len(sys.argv) > 4 else "", "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024",
           "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
