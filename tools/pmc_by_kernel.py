#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel (development aid).  usage: pmc_by_kernel.py <dir with *counter_collection.csv> """
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(lambda: collections.Counter())
n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
    print(k, "dispatches", n[k])
    for name, v in sorted(c.items()):
        print("   %-32s %.4g   per dispatch %.4g" % (name, v, v / max(1, n[k])))
