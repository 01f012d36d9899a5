"""Turns the raw rocprofv3 outputs of scripts/collect_profiles_r06.sh into the small files kept under profiles/ (tag r06)."""
import collections, csv, glob, json, os, shutil

G, P, tag = "gpurun_out", "profiles", "r06"
os.makedirs(P, exist_ok=True)


def newest(pat):
    f = sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:80]


# ---- headline workload: per-kernel stats of `bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline`
ks = newest(f"{G}/prof_{tag}/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, f"{P}/{tag}_bench_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(f"{P}/{tag}_bench_kernel_stats_top.txt", "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 6 --warmup 2 --headline-only (two-stream schedule = SpiderFreeInfer.submit: the decoder pass of request k beside the LLM pass of request k+1, enqueued from two host threads -- the kernels of both streams overlap under the profiler too; step time in r06_bench_under_rocprof.json; the averages include the two warm-up steps, whose passes run alone; compare with the serial file for every kernel ALONE on the chip) : total kernel time {tot / 1e6:.1f} ms\n")
        for r in rows[:25]:
            f.write(f"{short(r['Name']):80s} calls {int(float(r['Calls'])):7d} avg_us {float(r['AverageNs']) / 1e3:9.1f} "
                    f"total_ms {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['Percentage']):5.1f}%\n")
kss = newest(f"{G}/prof_{tag}_serial/**/*kernel_stats.csv")
if kss:
    rows = list(csv.DictReader(open(kss)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(f"{P}/{tag}_bench_serial_kernel_stats_top.txt", "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --headline-only --schedule serial (every kernel alone on the chip) : total kernel time {tot / 1e6:.1f} ms\n")
        for r in rows[:25]:
            f.write(f"{short(r['Name']):80s} calls {int(float(r['Calls'])):7d} avg_us {float(r['AverageNs']) / 1e3:9.1f} "
                    f"total_ms {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['Percentage']):5.1f}%\n")
for name in ("bench_line.json", "bench_under_rocprof.json", "bench_serial_under_rocprof.json"):
    src = f"{G}/{tag}_{name}"
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, f"{P}/{tag}_{name}")

# ---- UNet step alone: launches per step and time per kernel (22 evaluations: warm-up + capture warm-up + 20 replays)
ku = newest(f"{G}/prof_{tag}_unet/**/*kernel_stats.csv")
unet = {}
if ku:
    rows = list(csv.DictReader(open(ku)))
    ours = [r for r in rows if "at::native" not in r["Name"] and "rocclr" not in r["Name"]]
    # evaluations in the trace = calls of conv_in's kernel (exactly one per UNet evaluation): warm-up + capture warm-up + 20 replays
    # + the checksum step of scripts/prof_unet.py
    evals = float(next((r["Calls"] for r in ours if "conv_small_cin_kernel" in r["Name"]), 22))
    n_launch = sum(float(r["Calls"]) for r in ours) / evals
    t_ms = sum(float(r["TotalDurationNs"]) for r in ours) / 1e6 / evals
    log = open(f"{G}/{tag}_prof_unet.log").read() if os.path.exists(f"{G}/{tag}_prof_unet.log") else ""
    step = [l for l in log.splitlines() if l.startswith("unet step ms")]
    with open(f"{P}/{tag}_unet_step_stats.txt", "w") as f:
        f.write(f"UNET_DTYPE=f16 UNET_STREAM32=1 rocprofv3 --kernel-trace --stats -- python scripts/prof_unet.py 20   (SD-v1.5 UNet, f16 + fp32 residual stream, CFG batch 2, 64x64 latent; {evals:.0f} evaluations)\n")
        f.write(f"launches per UNet step: {n_launch:.0f}   kernel time per step: {t_ms:.3f} ms   wall (under the profiler): {step[-1] if step else 'n/a'}\n")
        for r in ours[:30]:
            f.write(f"{short(r['Name']):80s} n/step {float(r['Calls']) / evals:6.1f} avg_us {float(r['AverageNs']) / 1e3:8.1f} "
                    f"ms/step {float(r['TotalDurationNs']) / 1e6 / evals:7.3f}\n")
    unet = dict(launches_per_step=round(n_launch), kernel_ms_per_step=round(t_ms, 3))


# ---- zeroscope UNet3D step alone (2 x 16 frames at 40 x 72; scripts/bench_video.py 4: warm-up + 4 timed evaluations + VAE decode)
kv = newest(f"{G}/prof_{tag}_unet3d/**/*kernel_stats.csv")
if kv:
    rows = list(csv.DictReader(open(kv)))
    log = open(f"{G}/{tag}_prof_unet3d.log").read() if os.path.exists(f"{G}/{tag}_prof_unet3d.log") else ""
    step = [l for l in log.splitlines() if l.startswith("unet3d step ms")]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(f"{P}/{tag}_unet3d_step_stats.txt", "w") as f:
        f.write("rocprofv3 --kernel-trace --stats -- python scripts/bench_video.py 4   (zeroscope UNet3D, batch 2 x 16 frames at 40 x 72: 5 evaluations + the VAE decode of 16 frames)\n")
        f.write(f"total kernel time {tot / 1e6:.1f} ms   {step[-1] if step else ''}\n")
        for r in rows[:30]:
            f.write(f"{short(r['Name']):80s} calls {int(float(r['Calls'])):6d} avg_us {float(r['AverageNs']) / 1e3:8.1f} "
                    f"total_ms {float(r['TotalDurationNs']) / 1e6:8.2f} {float(r['Percentage']):5.1f}%\n")


def per_kernel(path, counter):
    f = newest(f"{path}/**/*counter_collection.csv")
    if not f:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


# ---- HBM bytes of the UNet kernels (FETCH_SIZE x2 on gfx950 for wide coalesced reads, WRITE_SIZE as read; KB -> bytes)
fe, wr = per_kernel(f"{G}/pmc_{tag}_unet_fetch", "FETCH_SIZE"), per_kernel(f"{G}/pmc_{tag}_unet_write", "WRITE_SIZE")
if fe:
    ours = [k for k in fe if not k.startswith("at::") and not k.startswith("void at::") and "rocblas" not in k and "rocclr" not in k]
    top = sorted(ours, key=lambda k: -fe[k][0] * fe[k][1])[:12]
    for must in ("gemm_dma_kernel<160, 4, true, 0, 64, true>", "attn_flash_pipe_kernel<64, true, 3>", "splitk_reduce_gn_kernel", "gemm_kernel<64, 64, false, 0, false, true>"):      # the bench.py roofline kernels
        top += [k for k in ours if k.startswith(must) and k not in top]
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python scripts/prof_unet.py 4; per-launch averages over all "
                   "launches of the kernel in the UNet step; FETCH_SIZE is reported in KB and counts 64 B per 128-B request on gfx950 for wide "
                   "coalesced streams (x2 applied in hbm_read_bytes), WRITE_SIZE is exact for 16-B-per-lane stores (MI355X_MICROARCH.md, HBM section)",
           "kernels": []}
    for k in top:
        out["kernels"].append(dict(kernel=k, launches=fe[k][0], FETCH_SIZE_avg_KB=round(fe[k][1], 1), hbm_read_bytes=int(fe[k][1] * 1024 * 2),
                                   WRITE_SIZE_avg_KB=round(wr.get(k, (0, 0.0))[1], 1), hbm_write_bytes=int(wr.get(k, (0, 0.0))[1] * 1024)))
    out.update(unet)
    json.dump(out, open(f"{P}/{tag}_pmc_unet_hbm.json", "w"), indent=1)

# ---- HBM bytes of the UNet3D step's kernels (temporal attention, GroupNorm passes, the big convs)
fe3, wr3 = per_kernel(f"{G}/pmc_{tag}_unet3d_fetch", "FETCH_SIZE"), per_kernel(f"{G}/pmc_{tag}_unet3d_write", "WRITE_SIZE")
if fe3:
    ours3 = [k for k in fe3 if not k.startswith("at::") and not k.startswith("void at::") and "rocblas" not in k and "rocclr" not in k]
    top3 = sorted(ours3, key=lambda k: -fe3[k][0] * fe3[k][1])[:12]
    top3 += [k for k in ours3 if k.startswith("attn_short_kernel") and k not in top3]
    out3 = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python scripts/bench_video.py 2 (zeroscope UNet3D, "
                    "2 x 16 frames at 40 x 72, bf16 engines); per-launch averages; FETCH_SIZE in KB, x2 gfx950 correction applied in hbm_read_bytes. "
                    "attn_short_kernel (temporal attention, one launch per sample): algorithmic 2880 x 5 sequence-heads x (3 x 16 x 64 + 16 x 64) x 2 B = "
                    "88.5 MB read + 29.5 MB written", "kernels": []}
    for k in top3:
        out3["kernels"].append(dict(kernel=k, launches=fe3[k][0], FETCH_SIZE_avg_KB=round(fe3[k][1], 1), hbm_read_bytes=int(fe3[k][1] * 1024 * 2),
                                    WRITE_SIZE_avg_KB=round(wr3.get(k, (0, 0.0))[1], 1), hbm_write_bytes=int(wr3.get(k, (0, 0.0))[1] * 1024)))
    json.dump(out3, open(f"{P}/{tag}_pmc_unet3d_hbm.json", "w"), indent=1)

# ---- decode weight streams
dec = {}
for key, d, kname in (("b1", "dec1", "gemv_kernel<1, 1, true, true"), ("b8", "dec8", "skinny_fm_kernel<1, 8, true>")):
    fz = per_kernel(f"{G}/pmc_{tag}_{d}_fetch", "FETCH_SIZE")
    hit = [k for k in fz if k.startswith(kname)]
    if hit:
        dec[key] = dict(kernel=hit[0], launches=fz[hit[0]][0], FETCH_SIZE_avg_KB=round(fz[hit[0]][1], 1),
                        hbm_read_bytes_corrected=int(fz[hit[0]][1] * 1024 * 2), algorithmic_bytes=271633408)
if dec:
    dec["note"] = ("decode gate/up projection (Qwen2.5-7B shapes): B = 1 weight-streaming GEMV, B = 8 fragment-major skinny MFMA GEMM; "
                   "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python scripts/prof_decode.py 12 / prof_decode_batch.py 8 12; x2 gfx950 correction applied")
    if "b1" in dec:
        dec["dominant_kernel"] = dict(dec["b1"], kernel="gemv_kernel<1,1,true,true> (decode gate/up + SwiGLU, Qwen2.5-7B shapes)")
    json.dump(dec, open(f"{P}/{tag}_pmc_decode_hbm.json", "w"), indent=1)

# ---- MFMA-busy
mf = {}
for c, pat in (("cross64", "attn_flash"), ("self64", "attn_flash"), ("self32", "attn_flash"), ("xattn", "xattn_fused"), ("conv", "gemm_dma_kernel")):
    f = newest(f"{G}/pmc_{tag}_mfma_{c}/**/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    if a.get("GRBM_GUI_ACTIVE", 0) > 0:     # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs = 1024 MFMA pipes
        a["mfma_busy_fraction"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8 * 1024)
    mf[{"xattn": "xattn_fused_64", "conv": "conv64_320"}.get(c, c)] = a
if mf:
    mf["note"] = ("SD-v1.5 shapes, CFG batch 2: self-attention 64^2 (4096 keys, d=40) and 32^2 (1024 keys, d=80), stand-alone 77-key cross-attention "
                  "(cross64), the fused cross-attention sub-block at the 64^2 site (xattn_fused_64: LayerNorm + to_q + attention + to_out in one "
                  "launch, scripts/pmc_xattn.py), the 3x3 conv 320->320 at 64^2; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA")
    json.dump(mf, open(f"{P}/{tag}_pmc_attn_mfma.json", "w"), indent=1)
# ---- round 5: the streaming conv's HBM traffic per launch
wf, ww = per_kernel(f"{G}/pmc_{tag}_ws_fetch", "FETCH_SIZE"), per_kernel(f"{G}/pmc_{tag}_ws_write", "WRITE_SIZE")
hit = [k for k in wf if k.startswith("wstream_kernel")]
if hit:
    k = hit[0]
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python scripts/exp/ws_conv_bench.py 1 0: SD-v1.5 8^2 conv 1280 -> 1280 at "
                       "CFG batch 2 (conv2 role: bias + temb + fp32 residual in / out), cold weights; per-launch averages; FETCH_SIZE x2 (gfx950, 16-B "
                       "streaming reads). Algorithmic: 29,491,200 B of weights + 327,680 B of activations; the K splits' partial slabs (5 x 655 KB, "
                       "written through with sc1 stores and read back by the combining blocks) are on top",
               "kernel": k, "launches": wf[k][0], "FETCH_SIZE_avg_KB": round(wf[k][1], 1), "hbm_read_bytes": int(wf[k][1] * 2048),
               "WRITE_SIZE_avg_KB": round(ww.get(k, (0, 0.0))[1], 1), "hbm_write_bytes": int(ww.get(k, (0, 0.0))[1] * 1024),
               "algorithmic_weight_bytes": 29491200}, open(f"{P}/{tag}_pmc_ws_conv.json", "w"), indent=1)
print("profiles:", sorted(x for x in os.listdir(P) if x.startswith(tag)))
