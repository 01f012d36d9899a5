"""Deterministic stand-ins for the checkpoint tokenizers (CLIPTokenizer / RobertaTokenizer): no vocabulary files travel with
the repo, and bench.py / the tests run on synthetic prompts. Same call surface as the pipelines use."""
import torch


class FakeTokenizer:
    """Hashes whitespace-separated words to ids; BOS=0, EOS/pad=2; same call surface the pipeline uses."""
    model_max_length = 77

    def __init__(self, vocab=400):
        self.vocab = vocab

    def _ids(self, text):
        return [0] + [3 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % (self.vocab - 3)) for w in text.split()] + [2]

    def __call__(self, prompts, padding="longest", max_length=None, truncation=False, return_tensors="pt"):
        if isinstance(prompts, str):
            prompts = [prompts]
        rows = [self._ids(p) for p in prompts]
        if truncation and max_length:
            rows = [r[:max_length - 1] + [2] if len(r) > max_length else r for r in rows]
        L = max_length if padding == "max_length" else max(len(r) for r in rows)
        rows = [r + [2] * (L - len(r)) for r in rows]
        class Out:
            pass
        o = Out()
        o.input_ids = torch.tensor(rows, dtype=torch.long)
        return o

    def batch_decode(self, ids):
        return [" ".join(f"w{int(t)}" for t in row if int(t) > 2) for row in ids]


class FakeRobertaTokenizer(FakeTokenizer):
    """RoBERTa-style stand-in: BOS=0, EOS=2, PAD=1, right padding, returns input_ids + attention_mask."""
    model_max_length = 32

    def __call__(self, prompts, padding="longest", max_length=None, truncation=False, return_tensors="pt"):
        if isinstance(prompts, str):
            prompts = [prompts]
        rows = [self._ids(p) for p in prompts]
        if truncation and max_length:
            rows = [r[:max_length - 1] + [2] if len(r) > max_length else r for r in rows]
        L = max_length if padding == "max_length" else max(len(r) for r in rows)
        class Out:
            pass
        o = Out()
        o.input_ids = torch.tensor([r + [1] * (L - len(r)) for r in rows], dtype=torch.long)
        o.attention_mask = torch.tensor([[1] * len(r) + [0] * (L - len(r)) for r in rows], dtype=torch.long)
        return o


