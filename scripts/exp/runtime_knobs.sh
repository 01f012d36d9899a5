#!/bin/bash
# ROCclr runtime knobs vs the two-stream step (and the one-stream step): which, if any, changes the per-launch cost
cd "$(dirname "$0")/../.."
run() { timeout -k 10 250 python bench.py --headline-only --steps 3 --warmup 1 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', d['ms_per_step'], d.get('overlap_last_step'))"; }
run base; run base "--schedule serial"
for kv in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 \
          DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 AMD_OPT_FLUSH=0 HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 \
          GPU_NUM_COMPUTE_RINGS=8 DEBUG_HIP_DYNAMIC_QUEUES=0 DEBUG_HIP_FORCE_ASYNC_QUEUE=1 GPU_FLUSH_ON_EXECUTION=1; do
  export $kv; run $kv; run $kv "--schedule serial"; unset ${kv%%=*}
done
