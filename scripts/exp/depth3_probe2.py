"""Probe (round 4): why does the decode loop run at half speed beside the UNet stream when it is started from a prefilled handle
(depth-3 pipelining) but not when it follows its own prompt pass (depth 2)? Variants of what stream L runs beside 41 UNet evaluations
on stream U (helper thread)."""
import sys, threading, time, torch
from spider_amd import ops
from spider_amd.llm import LlamaEngine, LLMConfig
from spider_amd.qwen_omni import QwenOmniThinker
from spider_amd.schedulers import PNDMScheduler
from spider_amd.unet import UNetConfig, UNetEngine
dev = torch.device("cuda:0")
cfg = LLMConfig.qwen25_7b()
llm = LlamaEngine.random_init(cfg, dev, max_batch=int(sys.argv[1]) if len(sys.argv) > 1 else 1, max_len=1536 + 136, seed=0)
th = QwenOmniThinker(llm)
ids = torch.randint(3, cfg.vocab, (1, 1536), device=dev)
am = torch.ones_like(ids)
kw = dict(max_new_tokens=128, eos_token_id=[], sync_every=128)
unet = UNetEngine.random_init(UNetConfig.sd15(), dev, seed=1, dtype=torch.float16, stream32=True)
enc = torch.randn(2, 77, 768, device=dev).half()
unet.prepare(PNDMScheduler().set_timesteps(40), enc)
x2 = ops.latent_to_nhwc(torch.randn(1, 4, 64, 64, device=dev), reps=2, dtype=torch.float16)
with ops.workspace_scope("image"):
    unet.step(x2, 0)
for cs in (0, 1):
    th.decode_finish(th.prefill_begin(ids, am, cache_set=cs, **kw))
torch.cuda.synchronize()
sL, sU = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)


def run(name, on_l, n_unet=41):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def u():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(sU), ops.workspace_scope("image"):
            ev[0].record(sU)
            for i in range(n_unet):
                unet.step(x2, i % 40)
            ev[1].record(sU)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = threading.Thread(target=u); t.start()
    with torch.cuda.stream(sL):
        ev[2].record(sL)
        on_l()
        ev[3].record(sL)
    t.join(); sU.synchronize(); sL.synchronize()
    print(f"{name:58s} wall {1e3 * (time.perf_counter() - t0):7.1f} ms   U {ev[0].elapsed_time(ev[1]):7.1f}   L {ev[2].elapsed_time(ev[3]):7.1f}", flush=True)


for rep in range(2):
    run("L: generate (prompt pass + decode loop)", lambda: th.generate(ids, am, **kw).cpu())
    h = th.prefill_begin(ids, am, cache_set=1, **kw); torch.cuda.synchronize()
    run("L: decode_finish of a handle prefilled before (set 1)", lambda: th.decode_finish(h).cpu())
    h = th.prefill_begin(ids, am, cache_set=0, **kw); torch.cuda.synchronize()
    run("L: decode_finish of a handle prefilled before (set 0)", lambda: th.decode_finish(h).cpu())
    run("L: prefill_begin + decode_finish (set 1)", lambda: th.decode_finish(th.prefill_begin(ids, am, cache_set=1, **kw)).cpu())
    h = th.prefill_begin(ids, am, cache_set=0, **kw); torch.cuda.synchronize()
    run("L: 30 ms sleep, then decode_finish (set 0)", lambda: (time.sleep(0.03), th.decode_finish(h).cpu()))
