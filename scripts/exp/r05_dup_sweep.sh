#!/bin/bash
# precise-mode cost against the thresholds of the split-once route: SPIDER_A32_DUP_GF (flops; 1000000 = a32 kernels everywhere) and
# SPIDER_A32_DUP_MIN_M (rows)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
for cfg in ${CFGS:-"1000000 0" "4 0" "4 600" "4 2100" "1 600"}; do
  set -- $cfg
  echo "== SPIDER_A32_DUP_GF=$1 SPIDER_A32_DUP_MIN_M=$2"
  SPIDER_A32_DUP_GF=$1 SPIDER_A32_DUP_MIN_M=$2 timeout -k 10 400 python3 scripts/exp/precise_cost.py 2>&1 | grep -v amdgpu.ids
done
