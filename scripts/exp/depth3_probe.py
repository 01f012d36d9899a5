"""Probe (round 4): decode-loop speed of LlamaEngine.prefill_begin / decode_finish per KV cache set, alone on the chip."""
import time, torch
from spider_amd.llm import LlamaEngine, LLMConfig
from spider_amd.qwen_omni import QwenOmniThinker
dev = torch.device("cuda:0")
cfg = LLMConfig.qwen25_7b()
llm = LlamaEngine.random_init(cfg, dev, max_batch=1, max_len=1536 + 136, seed=0)
th = QwenOmniThinker(llm)
ids = torch.randint(3, cfg.vocab, (1, 1536), device=dev)
kw = dict(max_new_tokens=128, eos_token_id=[], sync_every=128)
for cs in (0, 1, 0, 1):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h = th.prefill_begin(ids, torch.ones_like(ids), cache_set=cs, **kw)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out = th.decode_finish(h)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"set {cs} rep {rep}: prefill {1e3 * (t1 - t0):.1f} ms, decode loop {1e3 * (t2 - t1):.1f} ms", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
th.generate(ids, torch.ones_like(ids), **kw)
torch.cuda.synchronize(); print(f"generate: {1e3 * (time.perf_counter() - t0):.1f} ms")
