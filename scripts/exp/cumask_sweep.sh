#!/bin/bash
# two-stream schedule with the decoder (U) / LLM (L) stream restricted to a subset of the CUs
cd "$(dirname "$0")/../.."
run() { timeout -k 10 250 python bench.py --headline-only --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('overlap_last_step'))"; }
run base
for u in 64 96 128 160 192 224; do SPIDER_BENCH_CUMASK_U=$u run "U=$u"; done
for u in 128:2 64:4 64:2; do SPIDER_BENCH_CUMASK_U=$u run "U=$u"; done
SPIDER_BENCH_CUMASK_L=192 run "L=192"
SPIDER_BENCH_CUMASK_L=224 SPIDER_BENCH_CUMASK_U=128 run "L=224,U=128"
