#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
timeout -k 10 400 python -m pytest tests/test_hip_ops.py -k "test_conv" -x -q 2>&1 | tail -2
for m in 0 1; do
  echo "== SPIDER_CONV_HBITS=$m"; export SPIDER_CONV_HBITS=$m
  SHAPES=sdxl GEMM_CFGS="0:0,256:1" timeout -k 10 300 python scripts/bench_gemm.py 2>&1 | grep "^c\|shape"
  SHAPES=v3d GEMM_CFGS="0:0,256:1" timeout -k 10 300 python scripts/bench_gemm.py 2>&1 | grep "^vc"
  timeout -k 10 200 python scripts/bench_video.py 4 16 4 2>&1 | grep "unet3d step"
  timeout -k 10 300 python scripts/bench_story.py 20 2>&1 | grep "story 768"
done
