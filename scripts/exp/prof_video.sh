#!/bin/bash
# per-kernel profile of the zeroscope UNet3D step (2 x CAPS x 16 frames at 40 x 72)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD TMPDIR=/tmp
CAPS=${CAPS:-1}
rocprofv3 --kernel-trace --output-format csv --stats -d gpurun_out/prof_v3d -- python3 scripts/bench_video.py 4 16 $CAPS > gpurun_out/prof_v3d.log 2>&1
python3 scripts/show_stats.py "gpurun_out/prof_v3d/**/*kernel_stats.csv" 5 40
find gpurun_out/prof_v3d -name "*kernel_trace.csv" -delete
