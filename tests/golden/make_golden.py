"""Generates the golden fixtures in this directory by IMPORTING / EXECUTING the reference's own code
(read-only tree at /root/reference). Run in the build container only; the fixtures (data: inputs and
expected outputs) are committed, the reference source never is.

    python tests/golden/make_golden.py

Produces
  llama_ref_seed{0,1,2}.npz  tiny LlamaForCausalLM of spider/models/modeling_llama.py: weights (bf16 bits),
                             prompt ids, full-prompt logits, per-layer hidden states, 16 greedy tokens with a
                             manual KV-cache loop, per-step logits, top-2 margins
  llama_ref_d128_{0,1,2}.npz the same reference class at head_dim 128 (the HIP decode kernels' size): ids, logits, hidden states,
                             16 greedy tokens, margins; weights are regenerated from the recorded seed (checksum inside);
                             the 3 seeds of 3000 with the largest minimum top-2 margin (>= 0.1)
  llama_ops_ref.npz          per-op vectors: LlamaRMSNorm, apply_rotary_pos_emb, LlamaMLP, LlamaAttention
  routing_ref.json           SpiderDecoder.get_llm_text_res / get_llm_text_modality / generate (stub decoders)
                             and clean_prompt_array / extract_story_elements / extract_answer on 40+ strings
  routing_ref_fuzz.json      the same reference functions on 400 + 250 strings drawn from a small grammar (random tag order / nesting /
                             damage, regex metacharacters, unicode, <think> blocks, every clean_prompt_array fallback)
  moe_proj_ref.npz           the reference's own Mlp / TextFcLayerMoE bodies and preparing_output_embeds_infer on seeded inputs
  story_ref.npz              cal_attn_mask_xl masks (seeded) and SpatialAttnProcessor2_0 __call1__/__call2__ and a
                             7-step write-phase sequence with the coin flips recorded
"""
import ast
import importlib.util
import json
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))

from oracle.llama import LlamaCfg, LlamaOracle  # noqa: E402  (weights generator only)


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def extract_defs(path, names, ns):
    """exec selected top-level / class-level function or class definitions of a reference file in `ns`."""
    src = open(path).read()
    tree = ast.parse(src)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names and node.name not in found:
            found[node.name] = node
    for n in names:
        node = found[n]
        code = ast.get_source_segment(src, node)
        # de-indent methods
        lines = code.split("\n")
        ind = len(lines[0]) - len(lines[0].lstrip())
        first_col = node.col_offset
        lines = [lines[0]] + [l[first_col:] if l[:first_col].strip() == "" else l for l in lines[1:]]
        exec("\n".join(lines), ns)
    return ns


def bf16_bits(t):
    return t.bfloat16().view(torch.int16).numpy().copy()


# ----------------------------------------------------------------------------------------- LLM
def gen_llama():
    ml = load_by_path("ref_modeling_llama", f"{REF}/spider/models/modeling_llama.py")
    from transformers import LlamaConfig
    for seed in (0, 1, 2):
        cfg = LlamaCfg(hidden=64, layers=2, n_q=4, n_kv=4, head_dim=16, inter=128, vocab=97, rope_theta=10000.0,
                       eps=1e-6, max_pos=128)
        w = LlamaOracle.random_weights(cfg, seed=seed, std=0.35)
        hf = LlamaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                         num_hidden_layers=cfg.layers, num_attention_heads=cfg.n_q, rms_norm_eps=cfg.eps,
                         max_position_embeddings=cfg.max_pos, hidden_act="silu", pad_token_id=0, bos_token_id=1,
                         eos_token_id=2)
        model = ml.LlamaForCausalLM(hf).float().eval()
        missing, unexpected = model.load_state_dict(w, strict=False)
        assert not [m for m in missing if "rotary" not in m and "inv_freq" not in m], missing
        g = torch.Generator().manual_seed(100 + seed)
        ids = torch.randint(3, cfg.vocab, (2, 12), generator=g)
        with torch.no_grad():
            out = model(input_ids=ids, use_cache=True, output_hidden_states=True, return_dict=True)
            logits0 = out.logits
            hiddens = torch.stack(out.hidden_states, 0)  # [L+1, B, S, H]
            pkv = out.past_key_values
            toks, steps, margins = [], [], []
            cur = logits0[:, -1]
            for t in range(16):
                steps.append(cur)
                top2 = cur.topk(2, -1).values
                margins.append(top2[:, 0] - top2[:, 1])
                tok = cur.argmax(-1)
                toks.append(tok)
                pos = torch.full((2, 1), 12 + t, dtype=torch.long)
                o = model(input_ids=tok[:, None], past_key_values=pkv, position_ids=pos, use_cache=True, return_dict=True)
                pkv = o.past_key_values
                cur = o.logits[:, -1]
        np.savez_compressed(
            f"{OUT}/llama_ref_seed{seed}.npz",
            cfg=json.dumps(cfg.__dict__), names=np.array(list(w.keys())),
            **{f"w{i}": bf16_bits(v) for i, v in enumerate(w.values())},
            ids=ids.numpy(), logits0=logits0.numpy(), hiddens=hiddens.numpy(),
            tokens=torch.stack(toks, 1).numpy(), step_logits=torch.stack(steps, 1).numpy(),
            margins=torch.stack(margins, 1).numpy())
        print("llama seed", seed, "tokens", torch.stack(toks, 1)[0].tolist(), "min margin", float(torch.stack(margins).min()))

    # per-op vectors
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 5, 64, generator=g)
    norm = ml.LlamaRMSNorm(64, eps=1e-5)
    norm.weight.data = 1 + 0.1 * torch.randn(64, generator=g)
    q = torch.randn(2, 4, 6, 16, generator=g)
    k = torch.randn(2, 4, 6, 16, generator=g)
    rot = ml.LlamaRotaryEmbedding(16, max_position_embeddings=64)
    cos, sin = rot(q, seq_len=40)
    pos = torch.tensor([[3, 4, 5, 6, 7, 8], [30, 31, 32, 33, 34, 35]])
    qe, ke = ml.apply_rotary_pos_emb(q, k, cos, sin, pos)
    mlp = ml.LlamaMLP(64, 128, "silu")
    for p_ in mlp.parameters():
        p_.data = 0.2 * torch.randn(p_.shape, generator=g)
    np.savez_compressed(
        f"{OUT}/llama_ops_ref.npz",
        rms_x=x.numpy(), rms_w=norm.weight.data.numpy(), rms_y=norm(x).detach().numpy(), rms_eps=1e-5,
        rope_q=q.numpy(), rope_k=k.numpy(), rope_pos=pos.numpy(), rope_qe=qe.numpy(), rope_ke=ke.numpy(),
        mlp_x=x.numpy(), mlp_wg=mlp.gate_proj.weight.data.numpy(), mlp_wu=mlp.up_proj.weight.data.numpy(),
        mlp_wd=mlp.down_proj.weight.data.numpy(), mlp_y=mlp(x).detach().numpy())


# head_dim 128 (the size the HIP decode kernels are built for): the same reference class, fixtures small enough to commit because
# the weights are regenerated from the seed on both sides (LlamaOracle.random_weights; a checksum guards against RNG drift).
# Seeds are chosen so that every one of the 2 x 16 greedy decisions has a top-2 margin >= 0.1 of the reference's fp32 logits:
# the bf16 engine must then reproduce ALL token ids (north_star: routing bit-exact).
D128_CFG = dict(hidden=256, layers=2, n_q=2, n_kv=2, head_dim=128, inter=512, vocab=331, rope_theta=10000.0, eps=1e-6, max_pos=128)
D128_STD = 0.08
D128_PROMPT = 12


def gen_llama_d128(n_keep=3, seed_range=range(0, 3000)):
    ml = load_by_path("ref_modeling_llama", f"{REF}/spider/models/modeling_llama.py")
    from transformers import LlamaConfig
    cfg = LlamaCfg(**D128_CFG)
    hf = LlamaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                     num_hidden_layers=cfg.layers, num_attention_heads=cfg.n_q, rms_norm_eps=cfg.eps,
                     max_position_embeddings=cfg.max_pos, hidden_act="silu", pad_token_id=0, bos_token_id=1,
                     eos_token_id=2)
    runs = []
    for seed in seed_range:
        w = LlamaOracle.random_weights(cfg, seed=seed, std=D128_STD)
        model = ml.LlamaForCausalLM(hf).float().eval()
        missing, unexpected = model.load_state_dict(w, strict=False)
        assert not [m for m in missing if "rotary" not in m and "inv_freq" not in m], missing
        g = torch.Generator().manual_seed(100 + seed)
        ids = torch.randint(3, cfg.vocab, (2, D128_PROMPT), generator=g)
        with torch.no_grad():
            out = model(input_ids=ids, use_cache=True, output_hidden_states=True, return_dict=True)
            logits0, hiddens, pkv = out.logits, torch.stack(out.hidden_states, 0), out.past_key_values
            toks, steps, margins = [], [], []
            cur = logits0[:, -1]
            for t in range(16):
                steps.append(cur)
                top2 = cur.topk(2, -1).values
                margins.append(top2[:, 0] - top2[:, 1])
                tok = cur.argmax(-1)
                toks.append(tok)
                pos = torch.full((2, 1), D128_PROMPT + t, dtype=torch.long)
                o = model(input_ids=tok[:, None], past_key_values=pkv, position_ids=pos, use_cache=True, return_dict=True)
                pkv = o.past_key_values
                cur = o.logits[:, -1]
        mm = float(torch.stack(margins).min())
        runs.append((mm, seed, dict(ids=ids.numpy(), logits0=logits0.numpy(), hiddens=hiddens.numpy(),
                                    tokens=torch.stack(toks, 1).numpy(), step_logits=torch.stack(steps, 1).numpy(),
                                    margins=torch.stack(margins, 1).numpy(),
                                    wsum=np.float64(sum(float(v.double().abs().sum()) for v in w.values())))))
    runs.sort(key=lambda r: -r[0])
    for k, (mm, seed, d) in enumerate(runs[:n_keep]):
        assert mm >= 0.1, f"no seed with a healthy margin ({mm})"
        np.savez_compressed(f"{OUT}/llama_ref_d128_{k}.npz", cfg=json.dumps(D128_CFG), std=D128_STD, seed=seed, **d)
        print("llama d128 fixture", k, "seed", seed, "min margin", round(mm, 4), "tokens", d["tokens"][0].tolist())


# ----------------------------------------------------------------------------------------- routing
ROUTING_TEXTS = [
    "<IMAGE>apple</IMAGE><VIDEO>dog</VIDEO><AUDIO>cat</AUDIO>",
    "<MASK>apple</MASK>",
    "<IMAGE>a</IMAGE><VIDEO>b</VIDEO><AUDIO>c</AUDIO>",
    "plain text without any tag",
    "",
    "<IMAGE></IMAGE>",
    "<IMAGE>a red car</IMAGE> and <IMAGE>a blue car</IMAGE>",
    "<AUDIO>rain</AUDIO> first then <IMAGE>sun</IMAGE>",
    "<IMAGE>outer <IMAGE>inner</IMAGE> tail</IMAGE>",
    "<IMAGE>line one\nline two</IMAGE>",
    "<IMAGE>unterminated",
    "</IMAGE>reversed<IMAGE>",
    "<image>lowercase</image>",
    "<BOX>the cat</BOX><MASK>the dog</MASK>",
    "<VIDEO>a dog running</VIDEO><VIDEO>a cat sleeping</VIDEO><VIDEO>birds</VIDEO>",
    "Sure! Here is your picture: <IMAGE>a cozy cabin in the snow, 4k</IMAGE> Enjoy.",
    "<think>reasoning <IMAGE>ghost</IMAGE></think><IMAGE>real</IMAGE>",
    "<IMAGESTORY><GENERALPROMPT> 'a man with a black suit' </GENERALPROMPT> <PROMPTARRAY> ['wake up in the bed', 'have breakfast', 'work in the company', 'reading book in the home'] </PROMPTARRAY> <STYLENAME> 'Comic book' </STYLENAME></IMAGESTORY>",
    "<IMAGE>x</IMAGE><IMAGESTORY><GENERALPROMPT>g</GENERALPROMPT></IMAGESTORY>",
    "<AUDIO> spaces around </AUDIO>",
    "<IMAGE>caf\u00e9 \u732b \U0001F600</IMAGE>",
    "<IMAGE>a</IMAGE><IMAGE>b</IMAGE><AUDIO>c</AUDIO><IMAGE>d</IMAGE>",
    "<IMAGE>tab\there</IMAGE>",
    "<IMAGE>a</VIDEO><VIDEO>b</IMAGE>",
    "<IMAGE><VIDEO>nested other</VIDEO></IMAGE>",
    "<MASK>m1</MASK><MASK>m2</MASK>",
    "<BOX>b1</BOX>",
    "<IMAGE>.*?</IMAGE>",
    "<IMAGE>(group) [set] {brace} \\d+</IMAGE>",
    "<AUDIO>a</AUDIO>" * 5,
]

STORY_TEXTS = [
    "<GENERALPROMPT> 'a man with a black suit' </GENERALPROMPT> <PROMPTARRAY> ['wake up in the bed', 'have breakfast', 'work in the company', 'reading book in the home'] </PROMPTARRAY> <STYLENAME> 'Comic book' </STYLENAME>",
    "<think>hmm <GENERALPROMPT>wrong</GENERALPROMPT></think><GENERALPROMPT>right</GENERALPROMPT><PROMPTARRAY>[\"a\", \"b\"]</PROMPTARRAY><STYLENAME>Line art</STYLENAME>",
    "<GENERALPROMPT>first</GENERALPROMPT><GENERALPROMPT>last</GENERALPROMPT><PROMPTARRAY>['x']</PROMPTARRAY><PROMPTARRAY>['y', 'z']</PROMPTARRAY><STYLENAME>s1</STYLENAME><STYLENAME>s2</STYLENAME>",
    "<PROMPTARRAY>\n'wake up'\n'eat'\n'sleep'\n</PROMPTARRAY>",
    "<PROMPTARRAY>[wake up', 'eat', 'sleep]</PROMPTARRAY><GENERALPROMPT>\n multi\nline \n</GENERALPROMPT>",
    "<PROMPTARRAY>[\"wake up\", \"eat\"]</PROMPTARRAY>",
    "<PROMPTARRAY>[<b>'bold'</b>, 'plain']</PROMPTARRAY>",
    "<PROMPTARRAY>['', 'kept', 0, 5]</PROMPTARRAY>",
    "<PROMPTARRAY>   </PROMPTARRAY>",
    "<PROMPTARRAY>{'a': 1}</PROMPTARRAY>",
    "no tags at all",
    "<PROMPTARRAY>\"only one\"</PROMPTARRAY><STYLENAME>  'Japanese Anime'  </STYLENAME>",
    "<PROMPTARRAY>[\"json \\\"quoted\\\"\", \"two\"]</PROMPTARRAY>",
    "a</think>b</think><GENERALPROMPT>after first think only</GENERALPROMPT>",
]


def _routing_env():
    """the reference's SpiderDecoder with stub decoders + its story / answer parsers (shared by gen_routing and gen_routing_fuzz)"""
    # stub everything spider_decoder.py imports except the real registry
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Base(torch.nn.Module):
        def init_diffusion_model(self, *a, **k):
            return None
        def init_mask_decoder_sam(self, *a, **k):
            return None

    stub("spider"); stub("spider.models"); stub("spider.common")
    load_by_path("spider.common.registry", f"{REF}/spider/common/registry.py")
    stub("spider.common.utils")
    stub("spider.models.base_model", BaseModel=_Base)
    stub("spider.models.layers")
    stub("spider.models.custom_sd", StableDiffusionPipeline=object)
    stub("spider.models.custom_vd", TextToVideoSDPipeline=object)
    stub("spider.models.custom_ad", AudioLDMPipeline=object)
    stub("mmdet"); stub("mmdet.apis", init_detector=lambda *a, **k: None, inference_detector=None)
    torch.cuda.current_device = lambda: 0
    sd = load_by_path("spider.models.spider_decoder", f"{REF}/spider/models/spider_decoder.py")
    dec = sd.SpiderDecoder(diffusion_modules={}, mask_decoder_modules=None)
    calls = []
    def fake(mod, ret_none=False):
        def f(samples, **kw):
            calls.append([mod, samples["llm_text_res"][0]])
            return None if ret_none else [f"{mod}:{samples['llm_text_res'][0]}"]
        return f
    def fake_box(samples):
        calls.append(["BOX", samples["llm_text_res"][0]])
        return dict(outputs_bboxes=[["bb"]], outputs_label_names=[["ln"]], outputs_scores=[[0.9]])

    ns = {"re": __import__("re"), "ast": ast, "json": json}
    extract_defs(f"{REF}/spider_decoder_infer.py", ["clean_prompt_array", "extract_story_elements"], ns)
    extract_defs(f"{REF}/qwen2.5omni_spider_web.py", ["extract_answer"], ns)

    class _Self:
        pass
    self_ = _Self()
    self_.clean_prompt_array = lambda s: ns["clean_prompt_array"](self_, s)
    return dec, ns, self_, calls, fake, fake_box


def gen_routing():
    dec, ns, self_, calls, fake, fake_box = _routing_env()
    cases = []
    for none_mode in (False, True):
        for text in ROUTING_TEXTS:
            calls.clear()
            dec.decode_modality = dict(IMAGE=fake("IMAGE", none_mode), VIDEO=fake("VIDEO"), AUDIO=fake("AUDIO", none_mode),
                                       MASK=fake("MASK"), BOX=fake_box, IMAGESTORY=None)
            answers = []
            predictions = dict(IMAGE=[], VIDEO=[], AUDIO=[], MASK=[], BOX=dict(bboxes=[], label_names=[], scores=[]), IMAGESTORY=[])
            predictions_text = dict(IMAGE=[], VIDEO=[], AUDIO=[], MASK=[], BOX=[], IMAGESTORY=[], IMAGESTORY_prompts=[])
            samples = {"llm_text_all": [text]}
            a, p, pt = dec.generate(samples, answers, predictions, predictions_text)
            cases.append(dict(text=text, none_mode=none_mode, answers=a, predictions=p, predictions_text=pt,
                              calls=[list(c) for c in calls],
                              modality=dec.get_llm_text_modality(text, ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX"]),
                              res={m: dec.get_llm_text_res(text, m) for m in ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX", "IMAGESTORY"]}))
    story = []
    for text in STORY_TEXTS + ROUTING_TEXTS[17:19]:
        gp, pa, sn = ns["extract_story_elements"](self_, text)
        story.append(dict(text=text, general_prompt=gp, prompt_array=pa, style_name=sn,
                          answer=ns["extract_answer"](text)))
    json.dump(dict(cases=cases, story=story), open(f"{OUT}/routing_ref.json", "w"), indent=1, ensure_ascii=True)
    print("routing cases", len(cases), "story cases", len(story))


def _random_routing_text(r):
    """LLM-output-like strings from a small grammar: signal tags of the six modalities in random order and nesting, unterminated /
    reversed / lower-case / mismatched tags, captions with regex metacharacters, quotes, unicode, newlines and tabs, <think> blocks."""
    mods = ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX", "IMAGESTORY"]
    words = ["a red car", "dog", "the cat on a mat", "rain, heavy", "caf\u00e9 \u732b", "x", "", " ", "(a|b)*", "[set]", "$^.+?", "line\nbreak", "tab\there",
             "'quoted'", "\"double\"", "<b>bold</b>", "100% \\d+", "\U0001F600 smile", "A" * 40, "</think>", "<think>", "<IMAGE>", "</AUDIO>"]
    def caption(depth=0):
        parts = [r.choice(words) for _ in range(r.randint(0, 3))]
        if depth < 2 and r.random() < 0.25:
            parts.insert(r.randint(0, len(parts)), tag(depth + 1))
        return " ".join(parts)
    def tag(depth=0):
        m = r.choice(mods)
        k = r.random()
        if k < 0.70:
            return f"<{m}>{caption(depth)}</{m}>"
        if k < 0.76:
            return f"<{m}>{caption(depth)}"                        # unterminated
        if k < 0.82:
            return f"</{m}>{caption(depth)}<{m}>"                  # reversed
        if k < 0.88:
            return f"<{m.lower()}>{caption(depth)}</{m.lower()}>"   # wrong case
        if k < 0.94:
            return f"<{m}>{caption(depth)}</{r.choice(mods)}>"      # mismatched close
        return f"< {m} >{caption(depth)}</ {m}>"                    # spaces inside the brackets
    out = []
    for _ in range(r.randint(0, 6)):
        k = r.random()
        out.append(tag() if k < 0.6 else (r.choice(words) if k < 0.9 else f"<think>{caption()}</think>"))
    return r.choice(["", " ", "\n"]).join(out)


def _random_story_text(r):
    gens = ["'a man with a black suit'", "a cat", "  spaced  ", "", "\nmulti\nline\n", "\"dq\"", "<i>html</i>"]
    arrays = ["['wake up', 'eat', 'sleep']", "[\"a\", \"b\"]", "['one']", "[]", "'a'\n'b'\n'c'", "[wake up', 'eat]", "{'k': 1}", "['', 'kept', 0]",
              "[<b>'x'</b>, 'y']", "   ", "not a list", "['it\\'s', 'ok']", "[\"json \\\"q\\\"\", \"two\"]", "('t', 'u')", "['a',\n 'b',\n]", "[1, 2.5, None, True]"]
    styles = ["'Comic book'", "Line art", "  'Japanese Anime'  ", "", "(No style)", "\"Photographic\""]
    tags = [("GENERALPROMPT", gens), ("PROMPTARRAY", arrays), ("STYLENAME", styles)]
    parts = []
    for name, pool in tags:
        for _ in range(r.choice([0, 1, 1, 1, 2])):
            parts.append(f"<{name}>{r.choice([' ', '', chr(10)])}{r.choice(pool)}{r.choice([' ', ''])}</{name}>")
    r.shuffle(parts)
    text = r.choice(["", " ", "\n"]).join(parts)
    k = r.random()
    if k < 0.3:
        text = f"<think>plan <GENERALPROMPT>ghost</GENERALPROMPT> {r.choice(arrays)}</think>" + text
    elif k < 0.4:
        text = "a</think>b</think>" + text
    elif k < 0.5:
        text = f"<IMAGESTORY>{text}</IMAGESTORY>"
    return text


def gen_routing_fuzz(n_route=400, n_story=250, seed=2047):
    """A larger, randomly generated routing fixture through the SAME reference functions as gen_routing (routing_ref_fuzz.json)."""
    import random
    dec, ns, self_, calls, fake, fake_box = _routing_env()
    r = random.Random(seed)
    cases, seen = [], set()
    while len(cases) < n_route:
        text = _random_routing_text(r)
        none_mode = r.random() < 0.3
        if (text, none_mode) in seen:
            continue
        seen.add((text, none_mode))
        calls.clear()
        dec.decode_modality = dict(IMAGE=fake("IMAGE", none_mode), VIDEO=fake("VIDEO"), AUDIO=fake("AUDIO", none_mode),
                                   MASK=fake("MASK"), BOX=fake_box, IMAGESTORY=None)
        answers = []
        predictions = dict(IMAGE=[], VIDEO=[], AUDIO=[], MASK=[], BOX=dict(bboxes=[], label_names=[], scores=[]), IMAGESTORY=[])
        predictions_text = dict(IMAGE=[], VIDEO=[], AUDIO=[], MASK=[], BOX=[], IMAGESTORY=[], IMAGESTORY_prompts=[])
        a, p, pt = dec.generate({"llm_text_all": [text]}, answers, predictions, predictions_text)
        cases.append(dict(text=text, none_mode=none_mode, answers=a, predictions=p, predictions_text=pt, calls=[list(c) for c in calls],
                          modality=dec.get_llm_text_modality(text, ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX"]),
                          res={m: dec.get_llm_text_res(text, m) for m in ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX", "IMAGESTORY"]}))
    story, seen = [], set()
    while len(story) < n_story:
        text = _random_story_text(r)
        if text in seen:
            continue
        seen.add(text)
        try:
            gp, pa, sn = ns["extract_story_elements"](self_, text)
            err = None
        except Exception as ex:                      # the reference's own failure mode is part of the contract
            gp = pa = sn = None
            err = type(ex).__name__
        story.append(dict(text=text, general_prompt=gp, prompt_array=pa, style_name=sn, raises=err, answer=ns["extract_answer"](text)))
    json.dump(dict(cases=cases, story=story, seed=seed), open(f"{OUT}/routing_ref_fuzz.json", "w"), indent=0, ensure_ascii=True)
    print("routing fuzz cases", len(cases), "story fuzz cases", len(story), "story cases that raise in the reference:",
          sum(1 for s_ in story if s_["raises"]))


# ----------------------------------------------------------------------------------------- StoryDiffusion
class FakeAttn(torch.nn.Module):
    """Minimal stand-in for diffusers' Attention module fields used by SpatialAttnProcessor2_0."""
    def __init__(self, C, heads, g):
        super().__init__()
        self.heads = heads
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(C, C, bias=False)
        self.to_v = torch.nn.Linear(C, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Identity()])
        for p_ in self.parameters():
            p_.data = 0.3 * torch.randn(p_.shape, generator=g)
        self.spatial_norm = None
        self.group_norm = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
    def prepare_attention_mask(self, m, *a, **k):
        return m


def gen_story():
    import torch.nn.functional as F
    ns = {"torch": torch, "F": F, "random": random, "np": np}
    extract_defs(f"{REF}/StoryDiffusion/utils/gradio_utils.py", ["cal_attn_mask_xl"], ns)
    extract_defs(f"{REF}/StoryDiffusion/Comic_Generation.py", ["SpatialAttnProcessor2_0"], ns)
    cal = ns["cal_attn_mask_xl"]
    out = {}
    # masks for fixed seeds and sizes (device cpu, fp32 rand)
    for i, (seed, h, w) in enumerate([(0, 32, 32), (1, 64, 64), (2047, 128, 96)]):
        torch.manual_seed(seed)
        u1 = None
        st = torch.get_rng_state()
        m1, m4 = cal(5, 4, 0.5, 0.5, h, w, device="cpu", dtype=torch.float32)
        torch.set_rng_state(st)
        n1, n4 = (h // 32) * (w // 32), (h // 16) * (w // 16)
        r1 = torch.rand((1, 5 * n1), dtype=torch.float32)
        r4 = torch.rand((1, 5 * n4), dtype=torch.float32)
        out[f"mask{i}_hw"] = np.array([h, w]); out[f"mask{i}_rand1024"] = r1.numpy(); out[f"mask{i}_rand4096"] = r4.numpy()
        out[f"mask{i}_m1024"] = m1.numpy(); out[f"mask{i}_m4096"] = m4.numpy()

    # processor calls: N tokens per image, C channels, id_length 4, CFG batch 8
    g = torch.Generator().manual_seed(5)
    for tag, (N, Cc, heads, hh, ww) in {"a": (2, 64, 4, 32, 64), "b": (16, 64, 4, 64, 64)}.items():
        # choose height/width so that N == (h//32)*(w//32) -> the mask1024 branch; also test mask4096 branch via N2
        attn = FakeAttn(Cc, heads, g)
        proc = ns["SpatialAttnProcessor2_0"](id_length=4, device="cpu", dtype=torch.float32)
        hs = torch.randn(8, N, Cc, generator=g)
        torch.manual_seed(11)
        m1, m4 = cal(5, 4, 0.5, 0.5, hh, ww, device="cpu", dtype=torch.float32)
        n1 = (hh // 32) * (ww // 32)
        mk = m1 if N == n1 else m4
        msk = mk[: mk.shape[0] // 5 * 4, : mk.shape[0] // 5 * 4]
        with torch.no_grad():
            y1 = proc.__call1__(attn, hs, None, msk, None)
            y2 = proc.__call2__(attn, hs, None, None, None)
        out[f"proc{tag}_hs"] = hs.numpy(); out[f"proc{tag}_mask"] = msk.numpy()
        out[f"proc{tag}_y1"] = y1.numpy(); out[f"proc{tag}_y2"] = y2.numpy()
        for nme, p_ in attn.named_parameters():
            out[f"proc{tag}_{nme}"] = p_.data.numpy()
        out[f"proc{tag}_cfg"] = np.array([N, Cc, heads, hh, ww])

    # 7-step write-phase sequence through __call__ with module globals, 2 processors per step
    N, Cc, heads, hh, ww = 4, 64, 4, 32, 64   # nums_1024 = 2, nums_4096 = 8; use N = 8 -> mask4096 branch and N=2 -> 1024
    attn_a, attn_b = FakeAttn(Cc, heads, g), FakeAttn(Cc, heads, g)
    P = ns["SpatialAttnProcessor2_0"]
    pa, pb = P(id_length=4, device="cpu", dtype=torch.float32), P(id_length=4, device="cpu", dtype=torch.float32)
    ns.update(total_count=2, attn_count=0, cur_step=0, sa32=0.5, sa64=0.5, write=True, height=hh, width=ww)
    torch.manual_seed(2047); random.seed(2047)
    ns["mask1024"], ns["mask4096"] = cal(5, 4, 0.5, 0.5, hh, ww, device="cpu", dtype=torch.float32)
    seq_in, seq_out, coins, keep1024, keep4096 = [], [], [], [], []
    rs = random.getstate()
    for step in range(7):
        keep1024.append(ns["mask1024"][0].numpy().copy()); keep4096.append(ns["mask4096"][0].numpy().copy())
        xa = torch.randn(8, 2, Cc, generator=g)   # N = nums_1024 = 2
        xb = torch.randn(8, 8, Cc, generator=g)   # N = nums_4096 = 8
        # record the coin the processor will draw (only drawn when cur_step >= 5)
        st = random.getstate(); ca = random.random(); cb = random.random(); random.setstate(st)
        with torch.no_grad():
            ya = pa(attn_a, xa)
            yb = pb(attn_b, xb)
        coins.append([ca, cb] if step >= 5 else [-1.0, -1.0])
        seq_in.append((xa.numpy(), xb.numpy())); seq_out.append((ya.numpy(), yb.numpy()))
    out["seq_cfg"] = np.array([Cc, heads, hh, ww])
    out["seq_xa"] = np.stack([s[0] for s in seq_in]); out["seq_xb"] = np.stack([s[1] for s in seq_in])
    out["seq_ya"] = np.stack([s[0] for s in seq_out]); out["seq_yb"] = np.stack([s[1] for s in seq_out])
    out["seq_coins"] = np.array(coins); out["seq_keep1024"] = np.stack(keep1024); out["seq_keep4096"] = np.stack(keep4096)
    for tag, a in (("sa", attn_a), ("sb", attn_b)):
        for nme, p_ in a.named_parameters():
            out[f"seq_{tag}_{nme}"] = p_.data.numpy()
    np.savez_compressed(f"{OUT}/story_ref.npz", **out)
    print("story fixtures:", len(out), "arrays; final cur_step", ns["cur_step"])


def gen_qformer():
    """TextFcLayer, mode 'qformer' (spider/models/layers.py:76-98,125-139): the reference's own Q-Former building blocks
    `BertEmbeddings` and `BertLayer` (spider/models/Qformer.py:51-108,378-484, with BertSelfAttention / BertSelfOutput /
    BertAttention / BertIntermediate / BertOutput) executed on seeded inputs, driven the way `BertModel.forward` drives them for
    the call TextFcLayer makes (query_embeds only, encoder_hidden_states = fc(x), all-ones masks -> zero additive masks,
    query_length = number of query tokens; Qformer.py:804-966), between the module's own `fc` and `model` Linear layers.
    Qformer.py as a whole does not import under this image's transformers (its PreTrainedModel subclasses target v4.15: moved
    helpers, `init_weights` protocol), so the nn.Module classes that hold the arithmetic are executed and the mask / loop glue of
    BertModel.forward + BertEncoder.forward (no arithmetic beyond `(1 - mask) * -10000`) is restated in these few lines.
    `init_Qformer`'s bert-base-uncased config fields are used (hidden 768, 12 heads, gelu, eps 1e-12, 2 layers, cross-attention in
    every layer) with a narrower feed-forward to keep the run short."""
    import math
    from torch import nn
    from transformers.activations import ACT2FN
    from transformers.models.bert.configuration_bert import BertConfig
    from transformers.pytorch_utils import apply_chunking_to_forward, prune_linear_layer
    from oracle.moe_proj import QF_HIDDEN, QF_HEADS, QF_LAYERS, random_qformer_weights
    ns = dict(torch=torch, nn=nn, math=math, ACT2FN=ACT2FN, apply_chunking_to_forward=apply_chunking_to_forward,
              prune_linear_layer=prune_linear_layer, find_pruneable_heads_and_indices=None)     # the last two: prune_heads only, never called
    extract_defs(f"{REF}/spider/models/Qformer.py",
                 ["BertEmbeddings", "BertSelfAttention", "BertSelfOutput", "BertAttention", "BertIntermediate", "BertOutput", "BertLayer"], ns)
    in_dim, out_dim, n_query, inter = 64, 48, 7, 256
    cfg = BertConfig(hidden_size=QF_HIDDEN, num_attention_heads=QF_HEADS, intermediate_size=inter, hidden_act="gelu",
                     layer_norm_eps=1e-12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg.encoder_width, cfg.num_hidden_layers = QF_HIDDEN, QF_LAYERS                 # init_Qformer(num_output_tokens, hidden_dim = 768)
    cfg.add_cross_attention, cfg.cross_attention_freq, cfg.query_length = True, 1, n_query
    emb = ns["BertEmbeddings"](cfg).eval()
    layers = [ns["BertLayer"](cfg, l).eval() for l in range(QF_LAYERS)]
    fc, model = nn.Linear(in_dim, QF_HIDDEN), nn.Linear(QF_HIDDEN, out_dim)          # layers.py:79,91
    w = random_qformer_weights(in_dim, out_dim, n_query, inter=inter, seed=41)

    def load(mod, prefix):
        own = mod.state_dict()
        take = {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)}
        # everything of the checkpoint under this prefix must exist in the module; what the module has on top is the text branch
        # the reference's constructor deletes (word / position embeddings, layer.intermediate / layer.output) and buffers
        assert all(k in own for k in take), [k for k in take if k not in own]
        extra = [k for k in own if k not in take]
        assert all(any(t in k for t in ("word_embeddings", "position_embeddings", "position_ids", "intermediate.dense", "output.dense", "output.LayerNorm"))
                   and "_query" not in k and "attention" not in k for k in extra), extra
        own.update(take)
        mod.load_state_dict(own, strict=True)
    load(emb, "Qformer.bert.embeddings.")
    for l, lay in enumerate(layers):
        load(lay, f"Qformer.bert.encoder.layer.{l}.")
    load(fc, "fc.")
    load(model, "model.")
    g = torch.Generator().manual_seed(42)
    out = dict(in_dim=np.array(in_dim), out_dim=np.array(out_dim), n_query=np.array(n_query), inter=np.array(inter), seed=np.array(41),
               w_checksum=np.array(float(sum(v.double().abs().sum() for v in w.values()))))
    with torch.no_grad():
        for tag, (B, T) in {"a": (1, 1), "b": (2, 5), "c": (1, 9)}.items():
            x = torch.randn(B, T, in_dim, generator=g).bfloat16().float()
            enc = fc(x)                                                              # layers.py:126
            h = emb(query_embeds=w["query_tokens"].expand(B, -1, -1))                # Qformer.py:868-873
            self_mask = (1.0 - torch.ones(B, 1, 1, n_query)) * -10000.0              # get_extended_attention_mask of an all-ones mask
            enc_mask = (1.0 - torch.ones(B, 1, 1, T)) * -10000.0                     # invert_attention_mask(image_atts), layers.py:127
            for lay in layers:                                                       # BertEncoder.forward, Qformer.py:549-560
                h = lay(h, self_mask, None, enc, enc_mask, None, False, n_query)[0]
            out[f"{tag}_x"] = x.numpy(); out[f"{tag}_y"] = model(h).numpy()          # layers.py:139
    np.savez_compressed(os.path.join(OUT, "textfc_qformer_ref.npz"), **out)
    print("qformer fixture:", {k: v.shape for k, v in out.items() if k.endswith("_y")})


def gen_moe():
    """Trained-Spider output side: the reference's own TextFcLayerMoE / Mlp class bodies (spider/models/layers.py) and
    Spider.preparing_output_embeds_infer (spider/models/spider.py:1413-1463) executed here on seeded inputs."""
    from torch import nn
    from oracle.moe_proj import random_moe_weights
    ns = {"torch": torch, "nn": nn}
    extract_defs(f"{REF}/spider/models/layers.py", ["Mlp", "TextFcLayerMoE"], ns)
    in_dim = 64
    mods = {"IMAGE": dict(alignment_output_tokens=5, alignment_output_dim=64), "AUDIO": dict(alignment_output_tokens=1, alignment_output_dim=32)}
    m = ns["TextFcLayerMoE"](in_dim, mods, mode="moe_transformer", reconstruct_loss=False, device="cpu").eval()
    w = random_moe_weights(in_dim, mods, seed=31)
    missing, unexpected = m.load_state_dict(w, strict=True)
    g = torch.Generator().manual_seed(32)
    out = dict(in_dim=np.array(in_dim), seed=np.array(31), mods=json.dumps(mods))
    with torch.no_grad():
        # batch 1 only: `x_expert * routing_weights[:, :, expert]` (layers.py:265) broadcasts [B,T,512] with [B,1], which
        # is only well-formed for B == 1 -- the reference decodes one caption at a time (spider.py:1536-1541)
        for tag, (B, T, mod) in {"a": (1, 1, "IMAGE"), "b": (1, 7, "IMAGE"), "c": (1, 2, "AUDIO")}.items():
            x = torch.randn(B, T, in_dim, generator=g).bfloat16().float()
            out[f"{tag}_x"] = x.numpy(); out[f"{tag}_mod"] = np.array(mod); out[f"{tag}_y"] = m(x, modality=mod).numpy()

    # ---- preparing_output_embeds_infer with a stub `self`
    ns2 = {"torch": torch}
    extract_defs(f"{REF}/spider/models/spider.py", ["preparing_output_embeds_infer"], ns2)
    fn = ns2["preparing_output_embeds_infer"]
    H = 8
    BEG, END = {"IMAGE": 90, "AUDIO": 92}, {"IMAGE": 91, "AUDIO": 93}

    class Tok:
        def __call__(self, text, return_tensors="pt", add_special_tokens=False):
            mod = text.strip("</>")
            ids = torch.tensor([[END[mod] if text.startswith("</") else BEG[mod]]])
            class R:
                input_ids = ids
                def to(self, d): return self
            return R()

    class Self:
        device = "cpu"
        llama_tokenizer = Tok()
        using_lora = False
        output_alignment_modules = {"IMAGE": {"alignment_layer": [-1, 2]}, "AUDIO": {"alignment_layer": [-1]}}
        modality_tokens = {"IMAGE": 1, "AUDIO": 2}
        def embed_tokens(self, ids, using_lora=False):
            return ids.float().unsqueeze(-1).expand(*ids.shape, H) * 0.5
    # generated ids (sequences[0]); [1:] is what the function scans
    seq = [1, 7, 8, 90, 11, 12, 13, 50, 91, 9, 92, 21, 22, 60, 61, 93, 90, 31, 51, 91, 2]
    L = 4
    hs = tuple(tuple(torch.full((1, 1, H), float(step * 10 + layer)) for layer in range(L)) for step in range(len(seq)))
    class Outputs:
        sequences = torch.tensor([seq])
        hidden_states = hs
    cases = []
    for mod, mi in (("IMAGE", 0), ("IMAGE", 1), ("AUDIO", 0)):
        r = fn(Self(), {"TaskPrompt": [f"[{mod}]"]}, Outputs(), modality=mod, targets=None, modality_i=mi)
        cases.append(dict(modality=mod, modality_i=mi, hidden=[t.tolist() for t in r[1]], inputs=[t.tolist() for t in r[2]],
                          hidden_text=[t.tolist() for t in r[3]], inputs_text=[t.tolist() for t in r[4]]))
    out["capture"] = json.dumps(dict(seq=seq, H=H, L=L, begin=BEG, end=END, alignment_layer=Self.output_alignment_modules,
                                     modality_tokens=Self.modality_tokens, cases=cases))
    np.savez_compressed(f"{OUT}/moe_proj_ref.npz", **out)
    print("moe_proj_ref.npz", {k: v.shape for k, v in out.items() if k.endswith("_y")})


if __name__ == "__main__":
    torch.set_num_threads(4)
    if sys.argv[1:] == ["routing_fuzz"]:          # only the random routing fixture (the other generators are unchanged)
        gen_routing_fuzz()
        sys.exit(0)
    gen_llama()
    gen_llama_d128()
    gen_routing()
    gen_routing_fuzz()
    gen_story()
    gen_moe()
    gen_qformer()
