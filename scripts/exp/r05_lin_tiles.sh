#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
for t in 0 128 160 256 257; do
  SPIDER_GEMM_TILE=$t timeout -k 10 200 python3 scripts/exp/lin_tiles.py 2>&1 | grep tile
done
