#!/bin/bash
cd "$(dirname "$0")/../.."
run() { timeout -k 10 250 python bench.py --headline-only --steps 3 --warmup 1 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', d['ms_per_step'], d.get('overlap_last_step'))"; }
run base
for b in 2 4 8 16 32 64 128 256 512; do export DEBUG_HIP_GRAPH_BATCH_SIZE=$b; run batch=$b; done
