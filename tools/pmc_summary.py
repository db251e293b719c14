"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "psk" not in k:
                continue
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(f, k)
            for c, v in cs.items():
                vs = sorted(v)
                print("   %-24s n=%d median=%.6g mean=%.6g min=%.6g max=%.6g" % (c, len(v), vs[len(vs) // 2], sum(v) / len(v), min(v), max(v)))
