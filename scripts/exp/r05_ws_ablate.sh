#!/bin/bash
# ablation of the streaming conv (SPIDER_GEMM_DBG: 1 no compute, 2 no W stream, 4 no slab traffic) -- kernel-only device time per mode
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out
OUT=gpurun_out/r05_ws_ablate.txt
: > $OUT
for c in ${CASES:-0 2}; do
 for d in ${DBGS:-0 1 2 4 3 6 7}; do
  rm -rf /tmp/wsprof; mkdir -p /tmp/wsprof
  SPIDER_GEMM_DBG=$d TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv --stats -d /tmp/wsprof -- python3 scripts/exp/ws_conv_bench.py 1 $c > /dev/null 2>&1
  f=$(find /tmp/wsprof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$c" "$d" >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "wstream" in r["Name"]:
        print(f"case {sys.argv[2]} dbg {sys.argv[3]}: wstream avg_us {float(r['AverageNs']) / 1e3:8.2f}")
PY
 done
done
cat $OUT
