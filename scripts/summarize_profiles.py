"""Turns the raw rocprofv3 outputs of scripts/collect_profiles.sh into the small files kept under profiles/:
  <tag>_bench_kernel_stats.csv (+ _top.txt), <tag>_bench_line.json, <tag>_bench_under_rocprof.json,
  <tag>_pmc_decode_hbm.json (FETCH_SIZE / WRITE_SIZE per kernel, KB as the counter reports them; bench.py applies the x2
  gfx950 correction of MI355X_MICROARCH.md), <tag>_pmc_attn_mfma.json (MFMA-busy fraction of the UNet attention kernels)."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = "gpurun_out", "profiles"
os.makedirs(P, exist_ok=True)


def newest(pat):
    f = sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


ks = newest(f"{G}/prof_{tag}/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, f"{P}/{tag}_bench_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(f"{P}/{tag}_bench_kernel_stats_top.txt", "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 : total kernel time {tot / 1e6:.1f} ms\n")
        for r in rows[:25]:
            f.write(f"{r['Name'][:90]:90s} calls {int(float(r['Calls'])):7d} avg_us {float(r['AverageNs']) / 1e3:9.1f} "
                    f"total_ms {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['Percentage']):5.1f}%\n")
for name in ("bench_line.json", "bench_under_rocprof.json"):
    src = f"{G}/{tag}_{name}"
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, f"{P}/{tag}_{name}")


def per_kernel(path, counter):
    f = newest(f"{path}/**/*counter_collection.csv")
    if not f:
        return None
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        key = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
        agg.setdefault(key, []).append(float(r["Counter_Value"]))
    out = [dict(kernel=k[0], grid_size=int(k[1] or 0), launches=len(v), avg_KB=round(sum(v) / len(v), 1)) for k, v in agg.items()]
    return sorted(out, key=lambda d: -d["avg_KB"] * d["launches"])[:12]


hbm = {c: per_kernel(f"{G}/pmc_{tag}_{d}", c) for c, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"))}
if any(hbm.values()):
    dom = [d for d in (hbm["FETCH_SIZE"] or []) if d["kernel"].startswith("void gemv_kernel<1, 1, true, true>")]
    if dom:   # the roofline kernel of bench.py: KB -> bytes, x2 gfx950 correction
        hbm["dominant_kernel"] = dict(
            kernel="gemv_kernel<1,1,true,true> (decode gate/up + SwiGLU, Qwen2.5-7B shapes)", FETCH_SIZE_avg_KB=dom[0]["avg_KB"],
            note="gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM section)",
            hbm_read_bytes_corrected=int(dom[0]["avg_KB"] * 1024 * 2), algorithmic_bytes=271633408,
            command="rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python scripts/prof_decode.py 12 (WRITE_SIZE in a separate pass)")
    hbm["note"] = ("rocprofv3 --kernel-trace --pmc <counter> -- python scripts/prof_decode.py 12 (separate passes). Values are the "
                   "counter's KB per launch; on gfx950 FETCH_SIZE counts 64-byte units as 32 B: multiply by 2 (MI355X_MICROARCH.md).")
    json.dump(hbm, open(f"{P}/{tag}_pmc_decode_hbm.json", "w"), indent=1)

mf = {}
for c in ("cross64", "self64"):
    f = newest(f"{G}/pmc_{tag}_mfma_{c}/**/*counter_collection.csv")
    if not f:
        continue
    agg = {}
    for r in csv.DictReader(open(f)):
        if "attn_flash" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    if "GRBM_GUI_ACTIVE" in a and a["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs = 1024 MFMA pipes
        a["mfma_busy_fraction"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 1024)
    mf[c] = a
f = newest(f"{G}/pmc_{tag}_mfma_conv/**/*counter_collection.csv")
if f:
    agg = {}
    for r in csv.DictReader(open(f)):
        if "gemm_dma_kernel" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    if a.get("GRBM_GUI_ACTIVE", 0) > 0:
        a["mfma_busy_fraction"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 1024)
        a["note"] = "gemm_dma_kernel<160,3,CONV> on the UNet 3x3 conv 320->320 at the 64x64 latent, batch 2 (python scripts/pmc_gemm.py conv64_320)"
        mf["conv64_320"] = a
if mf:
    mf["note"] = ("SD-v1.5 UNet attention at 64x64 latent, CFG batch 2, 8 heads, d=40: cross (77 keys) and self (4096 keys); "
                  "python scripts/bench_attn.py <case> under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA")
    json.dump(mf, open(f"{P}/{tag}_pmc_attn_mfma.json", "w"), indent=1)
print("profiles:", sorted(os.listdir(P)))
