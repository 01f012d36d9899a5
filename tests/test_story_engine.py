"""GPU: StoryDiffusion consistent self-attention on the HIP kernels against (a) the reference-generated golden
sequence (tests/golden/story_ref.npz: SpatialAttnProcessor2_0 driven through 7 write-phase steps with recorded coin
flips) and (b) the fp32 oracle inside a tiny SDXL-like UNet in write and read mode; plus the story_generation API."""
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


class _Eng:
    """Minimal stand-in for UNetEngine: the hook only needs the fused qkv weight table."""
    def __init__(self, w):
        self.w = w


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("key_lists", [True, False])
def test_consistent_attention_matches_reference_sequence(dev, golden_dir, key_lists, dtype):
    from spider_amd.story import ConsistentSelfAttention, StoryState
    BF = torch.bfloat16 if dtype == "bf16" else torch.float16
    z = np.load(os.path.join(golden_dir, "story_ref.npz"))
    C, heads, hh, ww = [int(v) for v in z["seq_cfg"]]
    coins = iter([float(c) for row in z["seq_coins"] for c in row if c >= 0])
    torch.manual_seed(2047)   # CPU stream of the fixture generator (device="cpu" there)
    st = StoryState(total_count=2, height=hh, width=ww, coin=lambda: next(coins),
                    uniforms=lambda n: torch.rand((1, n), dtype=torch.float32).reshape(-1))
    st.regen_masks(dev)
    hook = ConsistentSelfAttention(st)
    hook.key_lists = key_lists       # visible-key lists (masked keys skipped) / keep-bits mask kernel: both pinned to the reference
    engs, outw = {}, {}
    for tag in ("sa", "sb"):
        t = lambda k: torch.from_numpy(z[f"seq_{tag}_{k}"])
        qkv = torch.cat([t("to_q.weight"), t("to_k.weight"), t("to_v.weight")], 0).to(BF)
        engs[tag] = _Eng({f"up_blocks.0.{tag}.attn1.qkv": qkv.to(dev)})
        outw[tag] = (t("to_out.0.weight"), t("to_out.0.bias"))
    for step in range(7):
        # fixture rows are mask row 0 = keep | own block 0; outside block 0 they must equal our keep vectors
        assert torch.equal(st.keep1024[2:], torch.from_numpy(z["seq_keep1024"][step])[2:])
        assert torch.equal(st.keep4096[8:], torch.from_numpy(z["seq_keep4096"][step])[8:])
        for tag, xk, yk in (("sa", "seq_xa", "seq_ya"), ("sb", "seq_xb", "seq_yb")):
            x = torch.from_numpy(z[xk][step]).to(BF)
            o = hook(engs[tag], f"up_blocks.0.{tag}.attn1", x.to(dev), heads).float().cpu()
            got = F.linear(o, outw[tag][0].to(BF).float(), outw[tag][1])
            ref = torch.from_numpy(z[yk][step])
            rel = float((got - ref).norm() / ref.norm())
            # 16-bit inputs / weights / P vs the reference's fp32 run (f16 measured 8x below bf16)
            assert rel < {"bf16": 2e-2, "f16": 3e-3}[dtype], (step, tag, rel)
    assert st.cur_step == 7


def _mk(sdxl_seed=5):
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.unet import UNetConfig
    ocfg = UNetCfg.tiny(True)
    w = random_unet_weights(ocfg, seed=sdxl_seed)
    return ocfg, w, UNetConfig(**ocfg.__dict__)


class _OracleHook:
    """Adapts oracle.story.ProcessorOracle to UNetOracle.attn_hook."""
    def __init__(self, st, w):
        from oracle import story as ostory
        self.st, self.w, self.os, self.procs = st, w, ostory, {}

    def wants(self, name):
        return name.startswith("up_blocks") and name.endswith("attn1")

    def __call__(self, unet, name, y, heads):
        w = self.w
        aw = self.os.AttnWeights(w[name + ".to_q.weight"], w[name + ".to_k.weight"], w[name + ".to_v.weight"],
                                 w[name + ".to_out.0.weight"], w[name + ".to_out.0.bias"], heads)
        return self.procs.setdefault(name, self.os.ProcessorOracle())(self.st, aw, y)


@pytest.mark.parametrize("dtype,stream32,precise", [("bf16", False, 0), ("f16", True, 0), ("f16", True, 2)])
def test_story_unet_write_then_read_matches_oracle(dev, dtype, stream32, precise):
    """bf16 with the 16-bit stream (round 2's mode), f16 with the fp32 residual stream (what init_story_generation loads), and the
    latter at precise=2 (the hook banks and projects the [hi | lo] rows of the fp32 LayerNorm): every branch of the processor -- plain
    steps, masked steps by the coin, write and read -- over 7 + 6 evaluations."""
    from oracle import story as ostory
    BF = torch.bfloat16 if dtype == "bf16" else torch.float16
    from oracle.unet import UNetOracle
    from spider_amd.story import ConsistentSelfAttention, StoryState
    from spider_amd.unet import UNetEngine
    ocfg, w, cfg = _mk()
    hh = ww = 64            # latent 8x8 -> attention at N = 16 (4x4) and N = 4 (2x2) = (h/16)^2 and (h/32)^2
    g = torch.Generator().manual_seed(1)
    coins_seq = [torch.rand(1, generator=g).item() for _ in range(200)]
    unis = [torch.rand(4000, generator=g) for _ in range(40)]

    def run(side, write_steps=7, read_steps=6):
        ci, ui = iter(coins_seq), iter(unis)
        hooks = dict(coin=lambda: next(ci), uniforms=lambda n: next(ui)[:n])
        outs = []
        if side == "oracle":
            unet = UNetOracle(ocfg, w)
            n_proc = sum(1 for k in w if k.startswith("up_blocks") and k.endswith(".attn1.to_q.weight"))
            st = ostory.StoryState(total_count=n_proc, height=hh, width=ww, **hooks)
            st.regen_masks()
            unet.attn_hook = _OracleHook(st, unet.w)
        else:
            unet = UNetEngine(cfg, w, dev, dtype=BF, stream32=stream32, precise=precise)
            st = StoryState(total_count=ConsistentSelfAttention.count_processors(unet), height=hh, width=ww, **hooks)
            st.regen_masks(dev)
            unet.self_attn_hook = ConsistentSelfAttention(st)
        gg = torch.Generator().manual_seed(2)
        for phase, B2, nsteps in (("write", 8, write_steps), ("read", 2, read_steps)):
            st.write, st.cur_step, st.attn_count = phase == "write", 0, 0
            enc = torch.randn(B2, 77, ocfg.cross_dim, generator=gg).bfloat16().float()
            added = dict(text_embeds=torch.randn(B2, 64, generator=gg).bfloat16().float(),
                         time_ids=torch.tensor([[hh, ww, 0, 0, hh, ww]] * B2, dtype=torch.float32))
            ts = torch.tensor([981 - 60 * i for i in range(nsteps)])
            if side != "oracle":
                unet.prepare(ts, enc.to(dev), added)
            for i, t in enumerate(ts):
                x = torch.randn(B2, 4, 8, 8, generator=gg).bfloat16().float()
                if side == "oracle":
                    outs.append(unet.forward(x, t, enc, added))
                else:
                    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
                    e = unet.step(xin if precise else xin.to(BF), i)
                    outs.append(e.permute(0, 3, 1, 2).float().cpu())
        return outs, st

    ref, st_o = run("oracle")
    got, st_g = run("hip")
    assert st_o.cur_step == st_g.cur_step == 6
    for i, (a, b) in enumerate(zip(got, ref)):
        rel = float((a - b).norm() / b.norm())
        print(f"MEASURED story_unet dtype={dtype} stream32={stream32} precise={precise} i={i} rel={rel:.5f}")
        # bf16: measured 1.58 - 2.14e-2 over the 13 write / read steps (+20 %); f16 + fp32 stream: 8x below; precise=2: inside 1e-3
        assert rel < (1.0e-3 if precise else {"bf16": 2.6e-2, "f16": 2.7e-3}[dtype]), (i, rel)      # f16 measured 1.60 - 2.21e-3


def _tiny_story_pipe(dev):
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.story import StableDiffusionXLPipeline
    from spider_amd.unet import UNetEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    from helpers import FakeTokenizer
    ocfg, w, cfg = _mk(9)
    c1 = CLIPCfg(400, 32, 2, 2, 64, 77)
    w1 = random_weights(clip_param_shapes(c1), seed=1)
    w2 = random_weights(clip_param_shapes(c1), seed=2)
    w2["text_projection.weight"] = torch.randn(64, 32) * 0.1
    vc = VAECfg(4, 3, (64, 64, 64, 128), 1, 32)
    pipe = StableDiffusionXLPipeline(UNetEngine(cfg, w, dev), VAEDecoderEngine(VAEConfig(**vc.__dict__), random_weights(vae_param_shapes(vc), seed=3), dev),
                                     CLIPTextEngine(CLIPTextConfig(**c1.__dict__), w1, dev), CLIPTextEngine(CLIPTextConfig(**c1.__dict__), w2, dev),
                                     FakeTokenizer(), FakeTokenizer(), DDIMScheduler())
    pipe.enable_freeu(0.6, 0.4, 1.1, 1.2)
    return pipe


def test_spider_story_free_infer_end_to_end(dev):
    """BASELINE configs[2] wiring (demo/inference_api.py:92-149): Question + ". " + system_prompt -> chat template -> native LLM
    generate -> decode -> extract_story_elements -> story_generation, containers filled as the reference fills them. The tiny
    random-weight LLM emits noise, so the tokenizer stand-in decodes it to a scripted response; what is checked is the contract."""
    from oracle.llama import LlamaCfg, LlamaOracle
    from spider_amd.llm import LlamaEngine, LLMConfig
    from spider_amd.spider_decoder import SpiderStoryFreeInfer
    ocfg = LlamaCfg(256, 2, 4, 2, 128, 512, 300, 10000.0, None, 1e-6, True, 256)
    llm = LlamaEngine(LLMConfig(**ocfg.__dict__), LlamaOracle.random_weights(ocfg, seed=3, std=0.08), dev, max_batch=1, max_len=96)
    seen = {}

    class Tok:
        def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True):
            seen["messages"], seen["agp"] = messages, add_generation_prompt
            return "<|user|>" + messages[0]["content"] + "<|assistant|>"

        def __call__(self, prompt, return_tensors="pt"):
            seen["prompt"] = prompt
            ids = torch.tensor([[3 + (ord(c) % 250) for c in prompt[:40]]])
            return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}

        def decode(self, ids, skip_special_tokens=True):
            seen["n_out"] = int(ids.shape[0])
            return ("<think>plan</think> <GENERALPROMPT> 'a red fox' </GENERALPROMPT> <PROMPTARRAY> ['wakes up', 'hunts', 'plays', "
                    "'sleeps', 'dreams'] </PROMPTARRAY> <STYLENAME> 'Comic book' </STYLENAME>")

    cfg = dict(model=dict(type="spider_free", name="spider_story_free_llama3", model_path="unused", system_prompt="SYS", max_context_len=12))
    infer = SpiderStoryFreeInfer(cfg, llm=llm, tokenizer=Tok(), story_pipe=_tiny_story_pipe(dev),
                                 story_kwargs=dict(height=64, width=64, num_steps=6, output_type="np"))
    answers, predictions, predictions_text = infer({"Question": ["a day of a fox"]})
    assert seen["messages"] == [{"role": "user", "content": "a day of a fox. SYS"}] and seen["agp"] is True
    assert seen["n_out"] == 40 + 12                                        # prompt + max_context_len new tokens, as HF returns them
    assert answers == predictions_text["IMAGESTORY"] and "<PROMPTARRAY>" in answers[0]
    assert predictions_text["IMAGESTORY_prompts"] == [["wakes up", "hunts", "plays", "sleeps", "dreams"]]
    assert len(predictions["IMAGESTORY"]) == 1 and len(predictions["IMAGESTORY"][0]) == 5      # 4 id panels + 1 real panel
    assert predictions["IMAGESTORY"][0][0].shape == (64, 64, 3) and predictions["IMAGE"] == []
    # a response without the three tags: no story, the reference's error path
    infer.tokenizer.decode = lambda ids, skip_special_tokens=True: "no tags here"
    a2, p2, t2 = infer({"Question": ["x"]})
    assert a2 == ["no tags here"] and p2["IMAGESTORY"] == [] and t2["IMAGESTORY_prompts"] == []


def test_story_generation_api(dev):
    """story_generation end to end on tiny engines: 4 id images + 1 real image, deterministic for a fixed seed."""
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.story import StableDiffusionXLPipeline, register_styles, story_generation
    register_styles({"Sketch": ("pencil sketch of {prompt}, cross hatching", "colour, photo")})
    from spider_amd.unet import UNetEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    from helpers import FakeTokenizer
    ocfg, w, cfg = _mk(9)     # cross_dim 64 = 32 + 32 (two tiny text encoders), pooled 64, time ids 6 x 32
    c1 = CLIPCfg(400, 32, 2, 2, 64, 77)
    w1 = random_weights(clip_param_shapes(c1), seed=1)
    w2 = random_weights(clip_param_shapes(c1), seed=2)
    w2["text_projection.weight"] = torch.randn(64, 32) * 0.1
    vc = VAECfg(4, 3, (64, 64, 64, 128), 1, 32)     # 4 levels -> x8, so latent = height / 8 as the N rule of the processor assumes
    pipe = StableDiffusionXLPipeline(UNetEngine(cfg, w, dev), VAEDecoderEngine(VAEConfig(**vc.__dict__), random_weights(vae_param_shapes(vc), seed=3), dev),
                                     CLIPTextEngine(CLIPTextConfig(**c1.__dict__), w1, dev), CLIPTextEngine(CLIPTextConfig(**c1.__dict__), w2, dev),
                                     FakeTokenizer(), FakeTokenizer(), DDIMScheduler())
    pipe.enable_freeu(0.6, 0.4, 1.1, 1.2)
    args = dict(general_prompt="a man with a black suit", prompt_array=["wake up", "have breakfast", "go to work", "read a book", "sleep"],
                style_name="Sketch", height=64, width=64, num_steps=7, output_type="np")
    a = story_generation(pipe, **args)
    b = story_generation(pipe, **args)
    assert len(a) == 5 and a[0].shape == (64, 64, 3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert pipe.unet.self_attn_hook is None


def test_keep_bits_packed_on_device_match_host_packing(dev):
    """The production mask path (device uniforms -> one ballot kernel) gives exactly the words of the host packing used by the
    reference-pinned tests, incl. the forced-False tail beyond id_length blocks and a ragged last word."""
    from spider_amd import ops
    from spider_amd.story import StoryState, pack_keep_bits
    for n, n_valid, thr in ((5 * 576, 4 * 576, 0.5), (5 * 100, 4 * 100, 0.3), (130, 130, 0.9)):
        u = torch.rand(n, generator=torch.Generator().manual_seed(n))
        keep = u < thr
        keep[n_valid:] = False
        got = ops.pack_keep_bits(u.to(dev), thr, n_valid).cpu()
        assert torch.equal(got, pack_keep_bits(keep)), (n, n_valid)
    st = StoryState(total_count=1, height=64, width=64)
    st.regen_masks(dev)
    assert st.keep1024 is None and st.keep_bits(True, 4 * 4, dev).is_cuda and st.keep_bits(False, 5 * 16, dev).dtype == torch.int64
