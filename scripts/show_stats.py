"""Print the top rows of a rocprofv3 kernel_stats.csv (optionally dividing totals by a per-step count)."""
import csv, glob, sys
pat, div = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(pat, recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: total {tot/1e6/div:.3f} ms per step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print(f"{r['Name'][:78]:78s} n/step {float(r['Calls'])/div:7.1f} avg_us {float(r['AverageNs'])/1e3:8.1f} ms/step {float(r['TotalDurationNs'])/1e6/div:7.3f} {float(r['Percentage']):5.1f}%")
