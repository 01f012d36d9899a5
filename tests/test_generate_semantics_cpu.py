"""CPU: the greedy-search bookkeeping of spider_amd.llm (per-row EOS, pad after EOS, stop when every row is finished,
StoppingCriteriaSub on sequence 0) against transformers' own `generate`, which is what the reference drives
(spider/models/spider.py:1492-1508, demo/inference_api.py:130, qwen2.5omni_spider_web.py:468)."""
import json
import os

import pytest
import torch

from spider_amd.llm import StoppingCriteriaSub, finalize_greedy, read_generation_config


def _tiny_hf(seed=0):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    cfg = LlamaConfig(vocab_size=97, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=128)
    m = LlamaForCausalLM(cfg).eval()
    for p in m.parameters():       # wide logits so that greedy streams are varied
        p.data.mul_(4.0)
    return m


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_finalize_greedy_equals_hf_generate(seed):
    m = _tiny_hf(seed)
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 97, (3, 7), generator=g)
    N = 24
    free = m.generate(ids, max_new_tokens=N, do_sample=False, eos_token_id=None, pad_token_id=0)[:, 7:]
    assert free.shape == (3, N)
    # choose EOS ids that occur at different steps in different rows (and one row possibly never)
    cands = [int(free[0, 5]), int(free[1, 11])]
    for pad in (None, 1):
        ref = m.generate(ids, max_new_tokens=N, do_sample=False, eos_token_id=cands, pad_token_id=pad if pad is not None else cands[0])[:, 7:]
        got, k, hit = finalize_greedy(free.clone(), cands, pad, None, None)
        assert got.shape == ref.shape and torch.equal(got, ref), (got, ref)
        assert k == ref.shape[1]
        # checked in blocks (sync_every = 4): the same answer as step-by-step checking
        res, checked = None, 0
        for n in range(4, N + 1, 4):
            tk, kk, h = finalize_greedy(free[:, :n].clone(), cands, pad, None, None, checked + 1)
            checked = n
            if h:
                res = tk
                break
        if res is None:
            res = finalize_greedy(free.clone(), cands, pad, None, None)[0]
        assert torch.equal(res, ref)


def test_stopping_criteria_sub_ends_the_batch_at_the_first_hit():
    m = _tiny_hf(3)
    ids = torch.randint(3, 97, (2, 6), generator=torch.Generator().manual_seed(9))
    N = 16
    free = m.generate(ids, max_new_tokens=N, do_sample=False, eos_token_id=None, pad_token_id=0)
    stop = [int(free[0, 6 + 4]), int(free[0, 6 + 5])]          # a two-token stop word of sequence 0
    from transformers import StoppingCriteriaList
    sc = StoppingCriteriaSub([stop])
    ref = m.generate(ids, max_new_tokens=N, do_sample=False, eos_token_id=None, pad_token_id=0,
                     stopping_criteria=StoppingCriteriaList([sc]))
    got, k, hit = finalize_greedy(free[:, 6:].clone(), None, None, [sc], ids)
    assert hit and torch.equal(torch.cat([ids, got], 1), ref)
    # found after the fact from a block check that starts past the hit's first token
    got2, k2, hit2 = finalize_greedy(free[:, 6:].clone(), None, None, [sc], ids, checked=3)
    assert hit2 and k2 == k


def test_generation_config_defaults(tmp_path):
    json.dump({"hidden_size": 8, "eos_token_id": 2, "thinker_config": {"text_config": {"pad_token_id": 9}}},
              open(tmp_path / "config.json", "w"))
    json.dump({"eos_token_id": [5, 6], "max_new_tokens": 77}, open(tmp_path / "generation_config.json", "w"))
    gc = read_generation_config(str(tmp_path))
    assert gc["eos_token_id"] == [5, 6] and gc["pad_token_id"] == 9 and gc["max_new_tokens"] == 77
    os.remove(tmp_path / "generation_config.json")
    assert read_generation_config(str(tmp_path))["eos_token_id"] == 2


def _hf_greedy_bookkeeping(free, eos, pad, criteria, prompt):
    """Step-by-step restatement of transformers' greedy `_sample` bookkeeping (the order of its loop body): pad finished rows, append,
    run the stopping criteria (per-row EOS + the caller's batch-level ones), stop when no row is unfinished."""
    B, n = free.shape
    unfinished = torch.ones(B, dtype=torch.long)
    if eos and pad is None:
        pad = eos[0]
    seq = torch.zeros(B, 0, dtype=torch.long)
    for t in range(n):
        nxt = free[:, t].clone()
        if eos:
            nxt = nxt * unfinished + pad * (1 - unfinished)
        seq = torch.cat([seq, nxt[:, None]], 1)
        done = torch.zeros(B, dtype=torch.bool)
        if eos:
            done |= torch.isin(nxt, torch.tensor(eos))
        full = seq if prompt is None else torch.cat([prompt, seq], 1)
        for sc in criteria or []:
            if bool(torch.as_tensor(sc(full, None)).all()):
                done |= True
        unfinished = unfinished & (~done).long()
        if int(unfinished.max()) == 0:
            return seq, t + 1, True
    return seq, n, False


@pytest.mark.parametrize("seed", range(40))
def test_finalize_greedy_random_blocks_against_stepwise_bookkeeping(seed):
    """finalize_greedy (vectorised, applied after the fact to blocks of tokens) against the step-by-step loop on random token blocks:
    several EOS ids, pad given or defaulted, rows that never finish, stop words on sequence 0 (also spanning block boundaries), and
    block-wise checking with `checked` exactly as LlamaEngine's sync_every loop drives it."""
    import random
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    B, n, V = rng.randint(1, 5), rng.randint(1, 30), rng.choice([4, 8, 30])
    free = torch.randint(0, V, (B, n), generator=g)
    eos = rng.choice([None, [rng.randrange(V)], sorted({rng.randrange(V) for _ in range(3)})])
    pad = rng.choice([None, rng.randrange(V), V + 3])
    prompt = rng.choice([None, torch.randint(0, V, (B, rng.randint(1, 4)), generator=g)])
    crit = None
    if rng.random() < 0.6:
        L = rng.randint(1, 3)
        s0 = rng.randrange(0, max(1, n - L + 1))
        word = free[0, s0:s0 + L].tolist() if rng.random() < 0.7 else [rng.randrange(V) for _ in range(L)]
        crit = [StoppingCriteriaSub([word])]
    want, kw, hw = _hf_greedy_bookkeeping(free, eos, pad, crit, prompt)
    got, k, hit = finalize_greedy(free.clone(), eos, pad, crit, prompt)
    assert (k, hit) == (kw, hw) and torch.equal(got, want), (seed, got, want)
    # block-wise, as the decode loop runs it: blocks of `se` tokens, criteria re-run only on the new prefixes
    se = rng.randint(1, 6)
    res, checked, kk = None, 0, None
    for m in range(se, n + se, se):
        m = min(m, n)
        tk, kk, h = finalize_greedy(free[:, :m].clone(), eos, pad, crit, prompt, checked + 1)
        checked = m
        if h or m == n:
            res, hh = tk, h
            break
    assert hh == hw and kk == kw and torch.equal(res, want), (seed, se, res, want)
