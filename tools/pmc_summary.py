#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter (FETCH_SIZE or WRITE_SIZE) per kernel from the counter_collection CSV.
usage: pmc_summary.py <dir-or-csv> <COUNTER>"""
import csv, glob, os, sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
tot, cnt = defaultdict(float), defaultdict(int)
for fn in files:
    with open(fn) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"].split("(")[0][:70]
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
print("# %s summed per kernel (raw counter units) from %s" % (counter, path))
for k in sorted(tot, key=lambda k: -tot[k]):
    print("%-72s dispatches %6d  sum %.6e" % (k, cnt[k], tot[k]))
