#!/bin/bash
# per-kernel profile of the SD-v1.5 VAE decode (one 512^2 image; 4 video frames)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD TMPDIR=/tmp
python3 scripts/exp/bench_vae.py 2>&1 | grep "ms"
rocprofv3 --kernel-trace --output-format csv --stats -d gpurun_out/prof_vae -- python3 scripts/exp/bench_vae.py > /dev/null 2>&1
python3 scripts/show_stats.py "gpurun_out/prof_vae/**/*kernel_stats.csv" 12 16
rm -rf gpurun_out/prof_vae
