#!/bin/bash
# per-kernel profile of the SDXL story step (4 panels, 768^2, CFG batch 8)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --stats -d gpurun_out/prof_story -- python3 scripts/bench_story.py 10 > gpurun_out/prof_story.log 2>&1
python3 scripts/show_stats.py "gpurun_out/prof_story/**/*kernel_stats.csv" 20 36
find gpurun_out/prof_story -name "*kernel_trace.csv" -delete
