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




class SyntheticOmniProcessor:
    """Stand-in for the checkpoint's Qwen2_5OmniProcessor (qwen2.5omni_spider_web.py:461-471) where no vocabulary files exist
    (bench.py, tests): same three methods `SpiderFreeInfer` calls. Random-init weights emit no signal tags, so `batch_decode` renders a
    row as one response line that carries exactly one caption per modality in `tags`, the caption built from the row's first
    GENERATED ids. Where the generated part starts is read from the row itself, like a chat template's assistant marker: `__call__`
    ends every prompt with the id ASSISTANT (1, which the word hash never produces) and `batch_decode` takes what follows its last
    occurrence -- no state carried from one request to the next, so requests may be decoded in another order than they were
    tokenised (SpiderFreeInfer depth 3 tokenises request k+2 before it decodes request k+1). Rows without the marker (prompts handed
    over as ready-made `input_ids`, bench.py) start at the fixed `prompt_len`."""
    ASSISTANT = 1

    def __init__(self, vocab=152064, tags=("IMAGE",), prompt_len=None, head=8):
        self.vocab, self.tags, self.prompt_len, self.head = vocab, tuple(tags), prompt_len, head
        self._tok = FakeTokenizer(vocab)

    def apply_chat_template(self, messages, add_generation_prompt=True, tokenize=False):
        parts = []
        for m in messages:
            c = m["content"]
            c = c if isinstance(c, str) else " ".join(str(p.get("text", "")) for p in c)
            parts.append(f"<|im_start|>{m['role']}\n{c}<|im_end|>")
        return "\n".join(parts) + ("\n<|im_start|>assistant\n" if add_generation_prompt else "")

    def __call__(self, text=None, audios=None, images=None, videos=None, return_tensors="pt", padding=True):
        rows = [self._tok._ids(t) + [self.ASSISTANT] for t in ([text] if isinstance(text, str) else list(text))]
        L = max(len(r) for r in rows)
        ids = torch.tensor([[2] * (L - len(r)) + r for r in rows], dtype=torch.long)          # LEFT padding, as batched generation needs
        mask = torch.tensor([[0] * (L - len(r)) + [1] * len(r) for r in rows], dtype=torch.long)
        self.prompt_len = L
        return {"input_ids": ids, "attention_mask": mask}

    def batch_decode(self, text_ids, skip_special_tokens=True, clean_up_tokenization_spaces=False):
        names = {"IMAGE": "scene", "AUDIO": "sound", "VIDEO": "clip"}
        out = []
        for row in text_ids:
            marks = (torch.as_tensor(row) == self.ASSISTANT).nonzero()
            S = int(marks[-1]) + 1 if marks.numel() else (self.prompt_len or 0)
            head = " ".join(str(int(t)) for t in row[S:S + self.head])
            out.append("system\nuser\nassistant\nSure. " + " ".join(f"<{m}>{names.get(m, 'item')} {head}</{m}>" for m in self.tags))
        return out
