#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -k "llm or generate or schedule or spider_model or qwen or decode" -x -q > gpurun_out/adv_t1.log 2>&1; echo EXIT $?; tail -3 gpurun_out/adv_t1.log
timeout -k 10 200 python scripts/prof_decode.py 128 2>&1 | tail -1
for i in 1 2; do timeout -k 10 250 python bench.py --headline-only --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('overlap_last_step'))"; done
