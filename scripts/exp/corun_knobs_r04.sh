#!/bin/bash
# Round-4 co-run experiments on the headline workload (bench.py --headline-only: two-stream schedule of SpiderFreeInfer.submit):
# which knob moves the LLM pass (the critical path) or the decoder pass when the two share the chip?
#   SPIDER_GEMV_XLDS_MAX=16384   the down projection's 38 KB activation copy in LDS -> read through L2 (fits beside 110-147 KB blocks)
#   SPIDER_BENCH_CUMASK_L/U      CU-masked streams
#   SPIDER_ATTN_INLINE=1         split-KV combine inside the attention launch (one launch less per layer)
#   SPIDER_GN_FUSE_IN=0          Transformer2DModel.norm as its own apply pass (2 launches more per transformer)
#   SPIDER_GN_PRODUCER=0         every GroupNorm runs its own statistics pass
run() { echo "== $*"; env "$@" python3 bench.py --steps 6 --warmup 3 --headline-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('overlap_last_step'), d['unet_step_ms'], d['roofline']['avg_launch_us'])"; }
run X=1
run SPIDER_GEMV_XLDS_MAX=16384
run SPIDER_ATTN_INLINE=1
run SPIDER_GN_FUSE_IN=0
run SPIDER_GN_PRODUCER=0 SPIDER_GN_FUSE_IN=0
run SPIDER_BENCH_CUMASK_L=224
run SPIDER_BENCH_CUMASK_L=192
run SPIDER_BENCH_CUMASK_L=192 SPIDER_BENCH_CUMASK_U=64:192
run X=2
