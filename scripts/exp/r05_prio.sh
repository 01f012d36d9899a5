#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
OUT=gpurun_out/r05_stream_prio.txt
: > $OUT
run() {
  tag="$1"; shift
  r=$(env "$@" timeout -k 10 300 python bench.py --headline-only --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['timed_steps']
print(t['step_wall_ms']['median'], t['step_wall_ms']['min'], t['step_wall_ms']['max'], t['per_step_median'].get('llm_pass_ms'), t['per_step_median'].get('decoder_pass_ms'))")
  echo "$tag: median/min/max step ms, llm span, decoder span = $r" | tee -a $OUT
}
for round in 1 2; do
  run "LLM 0, decoder -1 (default)" SPIDER_STREAM_PRIO=0,-1
  run "LLM -1, decoder 0" SPIDER_STREAM_PRIO=-1,0
  run "both 0" SPIDER_STREAM_PRIO=0,0
  run "LLM -1, decoder -1" SPIDER_STREAM_PRIO=-1,-1
done
