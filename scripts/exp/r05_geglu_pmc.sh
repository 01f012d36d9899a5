#!/bin/bash
# what the top-level GEGLU projection (92160 x 2x1280 x 320, LayerNorm-folded, gemm_p8_kernel<false,2,true>) is busy with: counter passes
# (kernel trace only, one small counter set per pass, as MI355X_MICROARCH.md prescribes)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD TMPDIR=/tmp LIN_ONLY="ln geglu 320"
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr " " "_" | cut -c1-40)
  rm -rf /tmp/pmc_geglu; mkdir -p /tmp/pmc_geglu
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc_geglu -- python3 scripts/exp/lin_tiles.py > /dev/null 2>&1
  f=$(find /tmp/pmc_geglu -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
if len(sys.argv) < 2 or not sys.argv[1]:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_p8_kernel" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
