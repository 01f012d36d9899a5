#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -k "attn_decode_fused" -x -q 2>&1 | tail -2
run() { timeout -k 10 250 python bench.py --headline-only --steps 3 --warmup 1 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', d['ms_per_step'], d.get('overlap_last_step'))"; }
run warm
for i in 0 1 0 1; do export SPIDER_ATTN_INLINE=$i; run inline=$i; run inline=$i "--schedule serial"; timeout -k 10 200 python scripts/prof_decode.py 128 2>&1 | tail -1; done
