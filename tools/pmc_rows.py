#!/usr/bin/env python3
"""pmc_rows.py -- per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more passes).
  python tools/pmc_rows.py <kernel substring> <csv> [<csv> ...]
Prints one line per counter: average value per dispatch of the matching kernel, and the dispatch count."""
import csv
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
key = sys.argv[1]
acc = defaultdict(list)
dur = []
for path in sys.argv[2:]:
    for row in csv.DictReader(open(path, newline="")):
        if key in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for name in sorted(acc):
    v = acc[name]
    print("%-28s %18.1f   (%d dispatches)" % (name, sum(v) / len(v), len(v)))
if dur:
    print("%-28s %18.1f us" % ("dispatch duration (avg)", sum(dur) / len(dur)))
