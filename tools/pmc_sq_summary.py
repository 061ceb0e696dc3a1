#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_sq.sh."""
import collections, csv, glob, sys
agg = collections.defaultdict(dict)
for f in sorted(glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv")):
    tmp = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tile_kernel" in k:
            tmp[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in tmp.items():
        for c, x in v.items():
            agg[k][c] = sum(x) / len(x)
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 4]:
    print(k)
    for c, v in agg[k].items():
        print("    %-34s %16.0f" % (c, v))
