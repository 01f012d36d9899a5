#!/bin/bash
# two-blocks-per-CU LDS-DMA linears + row-order epilogue (SPIDER_GEMM_NS2=1, default) against the previous dispatch (0): ms per
# evaluation of the three UNets (f16 + fp32 stream) and of SD-v1.5 at CFG batch 16; alternating processes
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD PRECISE_MODES=${PRECISE_MODES:-0} UNET_DTYPE=f16 UNET_STREAM32=1
for r in 1 2; do
  for v in 1 0; do
    echo "== SPIDER_GEMM_NS2=$v"
    SPIDER_GEMM_NS2=$v timeout -k 10 300 python3 scripts/exp/precise_cost.py 2>&1 | grep "ms per"
    SPIDER_GEMM_NS2=$v UNET_BATCH=16 timeout -k 10 200 python3 scripts/prof_unet.py 10 2>&1 | grep "unet step" | sed 's/^/sd15 CFG batch 16: /'
  done
done
