#!/bin/bash
# Experiment (round 4, runs at commit 9e591b5 -- the prefetcher was removed afterwards): SD-v1.5 UNet step (CFG batch 2, f16 + fp32
# stream) with / without an Infinity-Cache weight prefetcher forked beside the captured step; alternating runs on one box; knobs =
# fork granularity, blocks of the reader, window. Output: profiles/r04_weight_prefetch_ab.txt.
export UNET_DTYPE=f16 UNET_STREAM32=1 PYTHONPATH=.
run() { echo "== $*"; env "$@" timeout -k 10 200 python scripts/prof_unet.py 40 2>&1 | grep -E "unet step|checksum"; }
run SPIDER_WEIGHT_PREFETCH=0
run SPIDER_WEIGHT_PREFETCH=1 SPIDER_PREFETCH_GROUP_MB=16
run SPIDER_WEIGHT_PREFETCH=1 SPIDER_PREFETCH_GROUP_MB=32
run SPIDER_WEIGHT_PREFETCH=1 SPIDER_PREFETCH_GROUP_MB=64
run SPIDER_WEIGHT_PREFETCH=0
run SPIDER_WEIGHT_PREFETCH=1 SPIDER_PREFETCH_GROUP_MB=128 SPIDER_PREFETCH_WINDOW_MB=64
run SPIDER_WEIGHT_PREFETCH=1 SPIDER_PREFETCH_GROUP_MB=64 SPIDER_PREFETCH_BLOCKS=128
run SPIDER_WEIGHT_PREFETCH=1 SPIDER_PREFETCH_GROUP_MB=64 SPIDER_PREFETCH_BLOCKS=16
run SPIDER_WEIGHT_PREFETCH=1 SPIDER_PREFETCH_GROUP_MB=2000
run SPIDER_WEIGHT_PREFETCH=0
