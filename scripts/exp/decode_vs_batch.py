"""Decode step time against the number of rows decoded together (Qwen2.5-7B shapes, context 1536): the weight stream is shared by the
rows, so the step should cost what one row costs (2.85 ms) plus the rows' own KV reads and per-row arithmetic."""
import time
import torch
from spider_amd.llm import LlamaEngine, LLMConfig

dev = torch.device("cuda:0")
cfg = LLMConfig.qwen25_7b()
eng = LlamaEngine.random_init(cfg, dev, max_batch=8, max_len=1536 + 160, seed=0)
T = 96
print("rows  ms/step  tokens/s  (prefill ms)")
for B in (1, 2, 3, 4, 5, 6, 8):
    ids = torch.randint(3, cfg.vocab, (B, 1536), generator=torch.Generator().manual_seed(B))
    eng.generate(input_ids=ids, max_new_tokens=4, sync_every=4, eos_token_id=[])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.generate(input_ids=ids, max_new_tokens=2, eos_token_id=[]); torch.cuda.synchronize(); tp = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.generate(input_ids=ids, max_new_tokens=T, sync_every=T, eos_token_id=[]); torch.cuda.synchronize(); tg = time.perf_counter() - t0
    step = (tg - tp) / (T - 2)
    print(f"{B:4d} {step * 1e3:8.3f} {B / step:9.1f}   ({tp * 1e3:.1f})", flush=True)
