#!/bin/bash
# HBM fetch of the implicit-GEMM convs under the two K-tile walks (FETCH_SIZE in KB; x2 gfx950 correction NOT applied here)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD TMPDIR=/tmp
for m in 0 1; do for sh in conv64_320 conv48_640 conv3d_640; do
  export SPIDER_CONV_KCM=$m
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/kcm_pmc_${m}_$sh -- python3 scripts/pmc_gemm.py $sh > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/kcm_pmc_${m}_$sh/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "gemm" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("kcm=$m $sh", k, "launches", len(v), "FETCH_SIZE avg KB", round(sum(v[2:]) / max(1, len(v[2:])), 1))
PY
  rm -rf gpurun_out/kcm_pmc_${m}_$sh
done; done
