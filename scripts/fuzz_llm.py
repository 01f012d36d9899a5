"""Randomised differential check of the LLM engine (spider_amd.llm.LlamaEngine: prefill GEMMs + flash attention, decode GEMV graph,
batched decode on fragment-major weights, lm_head + argmax) against the fp32 oracle's greedy loop over random architectures and
requests: GQA ratios 1 ... 8, hidden sizes that are not powers of two, qkv bias, tied embeddings, llama3 rope scaling,
batches of 1 ... 8 rows with ragged LEFT-padded prompts (1 ... 90 tokens), eager against hipGraph decode. Token rule as in
tests/test_llm_engine.py: ids must agree up to the first position where the oracle's own top-2 margin is below bf16 resolution; the
step-0 logits must agree within the bf16 bounds. Not part of the suite; on the GPU box:

    PYTHONPATH=. python scripts/fuzz_llm.py [cases] [seed]          # default 40 cases"""
import random
import sys

import torch

from oracle.llama import LlamaCfg, LlamaOracle
from spider_amd.llm import LlamaEngine, LLMConfig

dev = torch.device("cuda:0")


def draw(r):
    """all random choices of one case (so that `--only i` reproduces case i without running the others)"""
    n_kv = r.choice([1, 2, 2, 4])
    group = r.choice([1, 2, 3, 4, 5, 6, 7, 8])          # the decode attention kernel's contract: GQA groups of 1 ... 8
    n_q = n_kv * group
    hidden = r.choice([256, 384, 448, 512, 640])
    inter = r.choice([256, 512, 704, 1024])
    vocab = r.choice([300, 331, 1000, 2049])
    layers = r.choice([1, 2, 3])
    rs = r.choice([None, None, dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                    original_max_position_embeddings=64)])
    theta = r.choice([10000.0, 500000.0, 1000000.0])
    ocfg = LlamaCfg(hidden, layers, n_q, n_kv, 128, inter, vocab, theta, rs, r.choice([1e-5, 1e-6]), r.random() < 0.4, 512, r.random() < 0.3)
    B = r.choice([1, 1, 2, 3, 5, 8])
    S = r.randint(1, 90)
    T = r.randint(1, 12)
    wseed = r.randint(0, 1 << 30)
    lens = [S] + [r.randint(1, S) for _ in range(B - 1)]           # ragged rows, LEFT-padded like a processor batch
    return dict(ocfg=ocfg, B=B, S=S, T=T, wseed=wseed, lens=lens, rs=rs, theta=theta)


def one_case(p, verbose=False):
    ocfg, B, S, T, wseed, lens, rs, theta = (p[k] for k in ("ocfg", "B", "S", "T", "wseed", "lens", "rs", "theta"))
    hidden, layers, n_q, n_kv, inter, vocab = ocfg.hidden, ocfg.layers, ocfg.n_q, ocfg.n_kv, ocfg.inter, ocfg.vocab
    desc = (f"llm hidden={hidden} layers={layers} n_q={n_q} n_kv={n_kv} inter={inter} vocab={vocab} rope={theta}/{'llama3' if rs else 'plain'} "
            f"bias={ocfg.qkv_bias} tied={ocfg.tie_embeddings} B={B} S={S} T={T} wseed={wseed}")
    # weight scale of the unit tests (0.08 at hidden 256), shrunk with the width so that attention scores stay O(1): at std 0.08 and
    # hidden 640 the softmax saturates and the fp32 / bf16 gap of a single row reaches 10 % (seed 0 case 66) without any token changing
    w = LlamaOracle.random_weights(ocfg, seed=wseed, std=0.08 * (256.0 / hidden) ** 0.5)
    g = torch.Generator().manual_seed(wseed)
    ids = torch.randint(3, vocab, (B, S), generator=g)
    am = torch.zeros(B, S, dtype=torch.long)
    for b, n in enumerate(lens):
        am[b, S - n:] = 1
        ids[b, :S - n] = 0
    ref_tok, ref_logits = LlamaOracle(ocfg, w).greedy(ids, T, attn_mask=am, return_logits=True)
    eng = LlamaEngine(LLMConfig(**ocfg.__dict__), w, dev, max_batch=B, max_len=S + T + 8)
    outs = {}
    for use_graph in (False, True):
        o = eng.generate(input_ids=ids, attention_mask=am, max_new_tokens=T, eos_token_id=[], return_dict_in_generate=True,
                         return_logits=True, use_graph=use_graph)
        outs[use_graph] = (o.sequences[:, S:].cpu(), o.logits.float().cpu())
    why = ""
    if not torch.equal(outs[False][0], outs[True][0]) or not torch.equal(outs[False][1], outs[True][1]):
        why = "graph decode and eager decode disagree"
    gen, lg = outs[True]
    e = 0.0
    for b in range(B):
        for t in range(T):
            # this step has the oracle's history: its logits are comparable. bf16 bounds of tests/test_llm_engine.py, widened for
            # up to 3 layers at hidden 640: relative L2 < 6 %, no logit off by more than 6 % of the logit range
            rl, gl = ref_logits[b, t], lg[b, t]
            rng = float(rl.abs().max())
            e_bt = float((gl - rl).norm() / rl.norm())
            e = max(e, e_bt)
            if verbose:
                print(f"  row {b} (prompt {lens[b]} of {S}) step {t}: rel {e_bt:.3e} max abs {float((gl - rl).abs().max()):.3f} range {rng:.2f} "
                      f"tok {int(gen[b, t])} / {int(ref_tok[b, t])}")
            if e_bt > 6e-2 or float((gl - rl).abs().max()) > 0.06 * rng:
                why = why or f"row {b} step {t}: logits off by rel {e_bt:.3e}, max abs {float((gl - rl).abs().max()):.3f} of range {rng:.2f}"
            if int(gen[b, t]) == int(ref_tok[b, t]):
                continue
            top2 = rl.topk(2).values
            margin = float(top2[0] - top2[1])
            if margin >= max(0.08, 0.12 * rng):                     # a decision well above the resolution of bf16 arithmetic
                why = why or f"row {b} step {t}: token {int(gen[b, t])} vs oracle {int(ref_tok[b, t])} at margin {margin:.3f} (logit range {rng:.2f})"
            break                                                   # histories differ from here on
    # a row answered alone gives the tokens it gets inside the batch (other kernels: weight-streaming GEMV vs batched / fragment-major),
    # wherever the oracle's decision is healthy
    if B > 1 and not why:
        solo = eng.generate(input_ids=ids[:1], attention_mask=am[:1], max_new_tokens=T, eos_token_id=[])[:, S:].cpu()
        for t in range(T):
            if int(solo[0, t]) != int(gen[0, t]):
                top2 = ref_logits[0, t].topk(2).values
                if int(gen[0, t]) == int(ref_tok[0, t]) and float(top2[0] - top2[1]) >= max(0.08, 0.12 * float(ref_logits[0, t].abs().max())):
                    why = f"row 0 alone decodes {int(solo[0, t])} at step {t}, inside the batch {int(gen[0, t])} (healthy margin)"
                break
    return desc, e, why


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else None
    r = random.Random(seed)
    bad, worst = 0, 0.0
    for i in range(n):
        p = draw(r)
        if only is not None and i != only:
            continue
        try:
            desc, e, why = one_case(p, verbose=only is not None)
        except Exception as ex:
            print(f"RAISED case {i}: {type(ex).__name__}: {str(ex).splitlines()[0][:300]}", flush=True)
            bad += 1
            continue
        worst = max(worst, e)
        if why:
            bad += 1
            print(f"FAIL case {i}: {desc}: {why}", flush=True)
    print(f"fuzz_llm: {n} cases, {bad} failed (seed {seed}); worst relative L2 of logits on matching histories {worst:.2e}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
