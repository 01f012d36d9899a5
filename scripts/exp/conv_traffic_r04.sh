#!/bin/bash
# Round 4: HBM bytes per launch of the 64^2 3x3 conv (8192 x 320 x 2880) in its three roles in the SD-v1.5 UNet step, against the
# bytes each role must move (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections as in MI355X_MICROARCH.md).
# Run on the GPU box from the repo root:  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && bash scripts/exp/conv_traffic_r04.sh
export PYTHONPATH=. UNET_DTYPE=f16
P="rocprofv3 --kernel-trace --output-format csv"
for c in conv64_320 conv64_320_gn conv64_320_res32; do
  $P --pmc FETCH_SIZE -d gpurun_out/pmc_conv_${c}_fetch -- python3 scripts/pmc_gemm.py $c > /dev/null 2>&1
  $P --pmc WRITE_SIZE -d gpurun_out/pmc_conv_${c}_write -- python3 scripts/pmc_gemm.py $c > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
A, W, C16, C32 = 8192 * 320 * 2, 320 * 2880 * 2, 8192 * 320 * 2, 8192 * 320 * 4
need = {"conv64_320": A + W + C16, "conv64_320_gn": A + W + C16, "conv64_320_res32": A + W + C32 + C16 + C32}
print("role                 kernel                                      read MB  write MB  total MB  algorithmic MB  ratio")
for c in need:
    tot = {}
    for kind, ctr, mul in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
        f = sorted(glob.glob(f"gpurun_out/pmc_conv_{c}_{kind}/**/*counter_collection.csv", recursive=True))[-1]
        v = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "gemm_" in r["Kernel_Name"]:
                v.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]) * mul)
        for k, xs in v.items():
            tot.setdefault(k, {})[kind] = sum(xs) / len(xs)
    for k, t in tot.items():
        rd, wr = t.get("fetch", 0) / 1e6, t.get("write", 0) / 1e6
        name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:42]
        print(f"{c:20s} {name:42s} {rd:8.1f} {wr:9.1f} {rd + wr:9.1f} {need[c] / 1e6:15.1f} {(rd + wr) * 1e6 / need[c]:6.2f}")
PY
find gpurun_out -name "*kernel_trace.csv" -path "*pmc_conv_*" -delete
find gpurun_out -name "*counter_collection.csv" -path "*pmc_conv_*" -delete
