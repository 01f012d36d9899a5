#!/bin/bash
# Round-end evidence run (on the GPU box, from the repo root):  bash scripts/collect_profiles.sh r01
# Produces under gpurun_out/: <tag>_bench_line.json (plain run), prof_<tag>/ (rocprofv3 --kernel-trace --stats of the same
# command), pmc_<tag>_fetch/ pmc_<tag>_write/ (HBM counters, separate passes, kernel-trace only, as MI355X_MICROARCH.md
# prescribes) and pmc_<tag>_mfma_{cross64,self64}/ (MFMA-busy of the UNet attention kernels). scripts/summarize_profiles.py
# turns them into the files committed under profiles/.
set -u
tag=${1:-r01}
export PYTHONPATH=. TMPDIR=/tmp
python3 bench.py --steps 3 --warmup 1 2> gpurun_out/${tag}_bench_err.log > gpurun_out/${tag}_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 3 --warmup 1 > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python3 scripts/prof_decode.py 12 > gpurun_out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- python3 scripts/prof_decode.py 12 > gpurun_out/pmc_${tag}_write.log 2>&1
for c in cross64 self64; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_${tag}_mfma_$c -- python3 scripts/bench_attn.py $c > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_${tag}_mfma_conv -- python3 scripts/pmc_gemm.py conv64_320 > /dev/null 2>&1
python3 scripts/summarize_profiles.py $tag
# gpurun merges at most 64 MiB back: keep the summaries, drop the raw traces (the per-launch kernel trace of the bench alone is ~50 MB)
mkdir -p gpurun_out/profiles_out && cp profiles/${tag}_* gpurun_out/profiles_out/
find gpurun_out -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/pmc_${tag}_mfma_cross64 gpurun_out/pmc_${tag}_mfma_self64 gpurun_out/pmc_${tag}_mfma_conv
