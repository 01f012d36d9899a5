#!/bin/bash
# per-kernel device times of the streaming conv and its reduce, per shape (rocprofv3 kernel trace, one shape per process)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out
OUT=gpurun_out/r05_ws_prof.txt
: > $OUT
for c in ${CASES:-0 2 6}; do
  rm -rf /tmp/wsprof; mkdir -p /tmp/wsprof
  TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv --stats -d /tmp/wsprof -- python3 scripts/exp/ws_conv_bench.py 1 $c 2>&1 | grep -E "^(c8|c16|up)" >> $OUT
  f=$(find /tmp/wsprof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $OUT <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("wstream", "splitk", "gemm_", "gn_")):
        print(f"    {n[:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:8.2f}")
PY
done
cat $OUT
