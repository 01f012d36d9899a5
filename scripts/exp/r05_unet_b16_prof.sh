#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD TMPDIR=/tmp UNET_DTYPE=f16 UNET_STREAM32=1
for b in ${BATCHES:-16}; do
  rm -rf /tmp/pp; mkdir -p /tmp/pp
  UNET_BATCH=$b rocprofv3 --kernel-trace --output-format csv --stats -d /tmp/pp -- python3 scripts/prof_unet.py 10 2>&1 | grep "unet step"
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $b <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = float(next((r["Calls"] for r in rows if "conv_small_cin" in r["Name"]), 13))
print(f"--- CFG batch {sys.argv[2]}  ({ev:.0f} evaluations)")
tot = 0
for r in rows[:26]:
    if "at::" in r["Name"]: continue
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:78]
    print(f"  {n:78s} n/step {float(r['Calls']) / ev:6.1f} avg_us {float(r['AverageNs']) / 1e3:8.1f} ms/step {float(r['TotalDurationNs']) / 1e6 / ev:7.3f}")
print("  kernel ms/step", sum(float(r["TotalDurationNs"]) for r in rows if "at::" not in r["Name"]) / 1e6 / ev)
PY
done
