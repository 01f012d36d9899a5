"""GPU time of the Qwen2.5-7B prompt pass (1536 tokens) with HIP events (round 4: SwiGLU in the gate/up GEMM's epilogue)."""
import torch
from spider_amd.llm import LlamaEngine, LLMConfig
dev = torch.device("cuda:0")
cfg = LLMConfig.qwen25_7b()
llm = LlamaEngine.random_init(cfg, dev, max_batch=1, max_len=1536 + 16, seed=0)
ids = torch.randint(3, cfg.vocab, (1, 1536), device=dev)
for _ in range(2):
    llm.prefill_begin(input_ids=ids, max_new_tokens=4)
ts = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    llm.prefill_begin(input_ids=ids, max_new_tokens=4)
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
print("prefill ms (events):", [round(t, 2) for t in ts])
