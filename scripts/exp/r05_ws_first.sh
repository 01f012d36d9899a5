#!/bin/bash
# round 5, first GPU pass of the weight-stationary streaming conv: parity tests, per-shape bench, whole UNet step A/B
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "wstream" > gpurun_out/r05_ws_tests.log 2>&1; echo TESTS EXIT $?; tail -5 gpurun_out/r05_ws_tests.log
timeout -k 10 300 python scripts/exp/ws_conv_bench.py 3 > gpurun_out/r05_ws_conv_bench.txt 2>&1; echo BENCH EXIT $?; cat gpurun_out/r05_ws_conv_bench.txt
for i in 1 2 3; do
  for ws in 0 128 512; do
    echo "SPIDER_WS_MAX_M=$ws: $(SPIDER_WS_MAX_M=$ws UNET_DTYPE=f16 UNET_STREAM32=1 timeout -k 10 200 python scripts/prof_unet.py 40 2>&1 | tail -2 | tr '\n' ' ')"
  done
done
