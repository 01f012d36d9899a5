#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD TMPDIR=/tmp UNET_DTYPE=f16 UNET_STREAM32=1
for p in 0 1; do
  rm -rf /tmp/pp; mkdir -p /tmp/pp
  UNET_PRECISE=$p rocprofv3 --kernel-trace --output-format csv --stats -d /tmp/pp -- python3 scripts/prof_unet.py 20 2>&1 | grep "unet step"
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $p <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = float(next((r["Calls"] for r in rows if "conv_small_cin" in r["Name"]), 23))
tot = 0
print(f"--- precise={sys.argv[2]}  ({ev:.0f} evaluations)")
for r in rows[:22]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:78]
    print(f"  {n:78s} n/step {float(r['Calls']) / ev:6.1f} avg_us {float(r['AverageNs']) / 1e3:8.1f} ms/step {float(r['TotalDurationNs']) / 1e6 / ev:7.3f}")
print("  launches/step", sum(float(r["Calls"]) for r in rows if "at::" not in r["Name"]) / ev)
PY
done
