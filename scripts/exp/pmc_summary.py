"""Average the PMC counters of one kernel from a rocprofv3 counter_collection.csv:  pmc_summary.py <dir> <kernel substring>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
agg = {}
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
