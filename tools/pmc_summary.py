#!/usr/bin/env python
"""Mean of every counter per kernel from a rocprofv3 --pmc ... --output-format csv directory."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    k = k.replace("void mv3d::", "")
    acc.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in acc.items():
    if filt in k:
        print(k[:70], "launches=%d" % len(next(iter(v.values()))), " ".join("%s=%.0f" % (c, sum(x) / len(x)) for c, x in v.items()))
