"""Profiling aid: batched decode (B sequences) at Qwen2.5-7B shapes."""
import sys, time, torch
from spider_amd.llm import LlamaEngine, LLMConfig
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 34
eng = LlamaEngine.random_init(LLMConfig.qwen25_7b(), dev, max_batch=B, max_len=1800)
ids = torch.randint(3, 150000, (B, 1536), device=dev)
eng.generate(input_ids=ids, max_new_tokens=4)
torch.cuda.synchronize()
t0 = time.perf_counter(); eng.generate(input_ids=ids, max_new_tokens=2, use_graph=False); torch.cuda.synchronize(); tp = time.perf_counter() - t0
t0 = time.perf_counter(); eng.generate(input_ids=ids, max_new_tokens=n, sync_every=n); torch.cuda.synchronize(); tg = time.perf_counter() - t0
print(f"B={B} prefill {tp*1e3:.1f} ms; decode {B*(n-2)/(tg-tp):.1f} tok/s ({(tg-tp)/(n-2)*1e3:.3f} ms/step)")
