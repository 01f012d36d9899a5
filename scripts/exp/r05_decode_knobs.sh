#!/bin/bash
# round 5: decode-attention knobs under the two-stream schedule, judged on the MEDIAN step of the timed region (bench.py timed_steps),
# variants alternating on one box
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
OUT=gpurun_out/r05_decode_knobs.txt
: > $OUT
run() {
  tag="$1"; shift
  r=$(env "$@" timeout -k 10 300 python bench.py --headline-only --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['timed_steps']
print(t['step_wall_ms']['median'], t['step_wall_ms']['min'], t['step_wall_ms']['max'], t['per_step_median'].get('llm_pass_ms'), t['per_step_median'].get('decoder_pass_ms'))")
  echo "$tag: median/min/max step ms, llm span, decoder span = $r" | tee -a $OUT
}
for round in 1 2 3; do
  run "default (nsplit 64, combine launch)" A=1
  run "nsplit 16" SPIDER_ATTN_NSPLIT=16
  run "nsplit 8" SPIDER_ATTN_NSPLIT=8
  run "inline combine, nsplit 64" SPIDER_ATTN_INLINE=1
  run "inline combine, nsplit 16" SPIDER_ATTN_INLINE=1 SPIDER_ATTN_NSPLIT=16
done
