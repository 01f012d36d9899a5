#!/bin/bash
# Round-6 evidence run (on the GPU box, from the repo root):  bash scripts/collect_profiles_r06.sh
# gpurun_out/: r06_bench_line.json (plain run), prof_r06/ (rocprofv3 --kernel-trace --stats of the headline workload),
# prof_r06_unet/ (kernel trace of the UNet step alone), pmc_r06_* (counter passes: kernel-trace only, one counter set per
# pass, as MI355X_MICROARCH.md prescribes). scripts/summarize_profiles_r06.py turns them into the files kept under profiles/.
set -u
export PYTHONPATH=. TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
python3 bench.py --steps 5 --warmup 2 2> gpurun_out/r06_bench_err.log > gpurun_out/r06_bench_line.json
# the headline command without its secondary timings: per-kernel averages = those of the timed region (two-stream schedule: the LLM
# kernels share the chip with the decoder pass) ...
$P --stats -d gpurun_out/prof_r06 -- python3 bench.py --steps 6 --warmup 2 --headline-only > gpurun_out/r06_bench_under_rocprof.json 2> gpurun_out/r06_prof.log
# ... and the same with the two passes of a response back to back on one stream (every kernel alone on the chip)
$P --stats -d gpurun_out/prof_r06_serial -- python3 bench.py --steps 3 --warmup 1 --headline-only --schedule serial > gpurun_out/r06_bench_serial_under_rocprof.json 2>> gpurun_out/r06_prof.log
export UNET_DTYPE=f16 UNET_STREAM32=1      # the engines as the pipelines / bench.py load them
$P --stats -d gpurun_out/prof_r06_unet -- python3 scripts/prof_unet.py 20 > gpurun_out/r06_prof_unet.log 2>&1
$P --stats -d gpurun_out/prof_r06_unet3d -- python3 scripts/bench_video.py 4 > gpurun_out/r06_prof_unet3d.log 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/pmc_r06_unet3d_fetch -- python3 scripts/bench_video.py 2 > /dev/null 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/pmc_r06_unet3d_write -- python3 scripts/bench_video.py 2 > /dev/null 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/pmc_r06_unet_fetch -- python3 scripts/prof_unet.py 4 > /dev/null 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/pmc_r06_unet_write -- python3 scripts/prof_unet.py 4 > /dev/null 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/pmc_r06_dec1_fetch -- python3 scripts/prof_decode.py 12 > /dev/null 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/pmc_r06_dec8_fetch -- python3 scripts/prof_decode_batch.py 8 12 > /dev/null 2>&1
M="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"
for c in cross64 self64 self32; do
  $P --pmc $M -d gpurun_out/pmc_r06_mfma_$c -- python3 scripts/bench_attn.py $c > /dev/null 2>&1
done
$P --pmc $M -d gpurun_out/pmc_r06_mfma_xattn -- python3 scripts/pmc_xattn.py > /dev/null 2>&1
$P --pmc $M -d gpurun_out/pmc_r06_mfma_conv -- python3 scripts/pmc_gemm.py conv64_320 > /dev/null 2>&1
# round 5 (kept in round 6): HBM traffic of the weight-stationary streaming conv (8^2, 1280 -> 1280: 29.5 MB of weights per launch)
$P --pmc FETCH_SIZE -d gpurun_out/pmc_r06_ws_fetch -- python3 scripts/exp/ws_conv_bench.py 1 0 > /dev/null 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/pmc_r06_ws_write -- python3 scripts/exp/ws_conv_bench.py 1 0 > /dev/null 2>&1
python3 scripts/summarize_profiles_r06.py
mkdir -p gpurun_out/profiles_out && cp profiles/r06_* gpurun_out/profiles_out/
# gpurun merges at most 64 MiB back: keep the summaries, drop the raw traces
find gpurun_out -name "*kernel_trace.csv" -delete
find gpurun_out -name "*counter_collection.csv" -delete
