#!/bin/bash
# round 6: decode-attention combine forms under the two-stream schedule (median step of bench.py's timed region) and alone (prof_decode),
# variants alternating on one box. Form 2 = in-launch combine without a fence (write-through partials, relaxed ticket, sc1 loads).
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
OUT=gpurun_out/r06_decode_knobs.txt
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
  run "default (nsplit 64, combine launch)" A=1
  run "form 2, nsplit 64" SPIDER_ATTN_INLINE=2
  run "form 2, nsplit 32" SPIDER_ATTN_INLINE=2 SPIDER_ATTN_NSPLIT=32
  run "form 2, nsplit 16" SPIDER_ATTN_INLINE=2 SPIDER_ATTN_NSPLIT=16
  run "combine launch, nsplit 32" SPIDER_ATTN_NSPLIT=32
done
