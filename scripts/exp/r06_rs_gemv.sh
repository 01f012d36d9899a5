#!/bin/bash
# round 6: row-streaming GEMV (SPIDER_GEMV_RS = resident workgroups per launch; 0 = short-block form) under the two-stream schedule
# (median step of bench.py's timed region) and alone (prof_decode), variants alternating on one box
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
OUT=gpurun_out/r06_rs_gemv.txt
: > $OUT
run() {
  tag="$1"; shift
  r=$(env "$@" timeout -k 10 300 python bench.py --headline-only --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['timed_steps']
print(t['step_wall_ms']['median'], t['step_wall_ms']['min'], t['step_wall_ms']['max'], t['per_step_median'].get('llm_pass_ms'), t['per_step_median'].get('decoder_pass_ms'))")
  a=$(env "$@" timeout -k 10 200 python scripts/prof_decode.py 128 2>&1 | tail -1)
  echo "$tag: median/min/max step ms, llm span, decoder span = $r | alone: $a" | tee -a $OUT
}
for round in 1 2; do
  run "short-block form (default)" SPIDER_GEMV_RS=0
  for n in ${RSB:-256 512 1024}; do
    run "row-streaming form, $n workgroups" SPIDER_GEMV_RS=$n
  done
done
