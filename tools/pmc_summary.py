#!/usr/bin/env python3
"""pmc_summary.py -- HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950: MI355X_MICROARCH.md, PMC slots), corrected as that guide's HBM section prescribes: FETCH_SIZE and
WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reports half the bytes of a 16 B/lane coalesced stream, so
read bytes = 2 * FETCH_SIZE * 1024.

  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <algorithmic bytes> "<command>" > profiles/rNN_pmc_summary.json
Only kernels of this library (zr::...) are kept.  bench.py reads the newest summary for `roofline.traffic`."""
import csv
import json
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def per_kernel(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path, newline="")):
        name = row["Kernel_Name"]
        if row["Counter_Name"] == counter and "zr::" in name:
            short = name[name.index("zr::"):]
            short = short.split("(")[0]
            acc[short].append(float(row["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
doc = {"command": sys.argv[4] if len(sys.argv) > 4 else "",
       "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reports half the bytes of a 16 B/lane "
                "coalesced stream (MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024",
       "kernels": {}, "bytes_per_launch_algorithmic": int(sys.argv[3])}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    e = {}
    if f:
        e["FETCH_SIZE_avg_KiB"] = sum(f) / len(f)
        e["dispatches_fetch"] = len(f)
    if w:
        e["WRITE_SIZE_avg_KiB"] = sum(w) / len(w)
        e["dispatches_write"] = len(w)
    if f and w:
        e["hbm_bytes_per_launch_corrected"] = int(2 * 1024 * e["FETCH_SIZE_avg_KiB"] + 1024 * e["WRITE_SIZE_avg_KiB"])
    doc["kernels"][k] = e
print(json.dumps(doc, indent=1))
