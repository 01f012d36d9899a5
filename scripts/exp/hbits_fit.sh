#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
timeout -k 10 400 python -m pytest tests/test_hip_ops.py -k "test_conv" -x -q 2>&1 | tail -2
SHAPES=v3d GEMM_CFGS="0:0,128:1,160:1,256:1" timeout -k 10 400 python scripts/bench_gemm.py 2>&1 | grep "^v\|shape"
SHAPES=sdxl GEMM_CFGS="0:0,128:1,160:1,256:1" timeout -k 10 400 python scripts/bench_gemm.py 2>&1 | grep "^c\|^x\|shape"
SHAPES=deep GEMM_CFGS="0:0" timeout -k 10 300 python scripts/bench_gemm.py 2>&1 | grep "^c"
GEMM_CFGS="0:0" timeout -k 10 300 python scripts/bench_gemm.py 2>&1 | grep "^c"
for m in 0 1; do export SPIDER_CONV_HBITS=$m; echo "HBITS=$m"
  timeout -k 10 200 python scripts/bench_video.py 6 16 1 2>&1 | grep "unet3d step"
  timeout -k 10 200 python scripts/bench_video.py 4 16 4 2>&1 | grep "unet3d step"
  timeout -k 10 300 python scripts/bench_story.py 20 2>&1 | grep "story 768"
  UNET_DTYPE=f16 UNET_STREAM32=1 timeout -k 10 200 python scripts/prof_unet.py 20 2>&1 | tail -1
done
