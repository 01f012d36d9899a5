"""Join scripts/exp/unet_calls.py's call log with the rocprofv3 kernel trace of the same process: walk both from the end
(a call = one main kernel, optionally followed by splitk_reduce; groupnorm = gn_stats + gn_apply or gn_small), print
per-call device time in launch order and totals per (entry point, shape)."""
import csv, glob, json, sys
from collections import defaultdict
calls = json.load(open(sys.argv[1]))
trace = sorted(glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ours = ("gemm", "splitk", "attn_", "xattn", "gn_", "concat", "conv_small", "act_kernel", "geglu", "layernorm", "latent", "add_kernel",
        "upsample", "lincomb", "cfg", "rope", "swiglu", "ln_")
ks = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
i = len(ks) - 1
out = []
def pop():
    global i
    while i >= 0 and not any(o in ks[i][0] for o in ours):
        i -= 1
    k = ks[i]; i -= 1
    return k
for c in reversed(calls):
    name = c[0]
    k = pop(); t = k[1]; names = [k[0].split("(")[0].replace("void ", "")]
    if "splitk_reduce" in k[0] or "gn_apply" in k[0]:
        k2 = pop(); t += k2[1]; names.insert(0, k2[0].split("(")[0])
    if "gemm_ln" in name and i >= 0 and "ln_row_stats" in ks[i][0]:      # the 256^2 LN form: row statistics launched first
        k3 = pop(); t += k3[1]; names.insert(0, k3[0].split("(")[0])
    out.append((name, [a for a in c[1:] if a is not None], t, names))
out.reverse()
agg = defaultdict(lambda: [0, 0.0, None])
tot = 0.0
with open(sys.argv[3], "w") as f:
    for name, a, t, names in out:
        f.write(f"{t:8.1f} us  {name:34s} {str(a):60s} {' + '.join(names)}\n")
        key = (name, tuple(a)); agg[key][0] += 1; agg[key][1] += t; agg[key][2] = names; tot += t
    f.write(f"\ntotal {tot:.1f} us over {len(out)} calls\n\nby (call, ints), sorted by total time:\n")
    for key, (n, t, names) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{t:8.1f} us  n={n:3d} avg {t / n:7.1f}  {key[0]:30s} {str(list(key[1])):56s} {names[0][:60]}\n")
print("joined", len(out), "calls, total", tot, "us")
