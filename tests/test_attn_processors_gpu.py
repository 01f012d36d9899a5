"""GPU: the diffusers-style processor registry of UNetEngine (spider_amd/attn_processors.py) used the way
StoryDiffusion/Comic_Generation.py:353-371 uses `unet.attn_processors` / `unet.set_attn_processor`: a user-written processor
(plain SDPA on torch, the shape of gradio_utils.py:400-472's AttnProcessor2_0) plugged into every self-attention gives the native
kernels' result; default processors keep the native path; the error behaviour of diffusers' dict form."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


class AttnProcessor:          # the reference's default processor: recognised by its class name, native kernels stay in place
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        raise AssertionError("default processors are not called")


class SdpaProcessor:
    """what a user would write against diffusers' Attention module"""
    calls = 0

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        SdpaProcessor.calls += 1
        residual = hidden_states
        b, n, c = hidden_states.shape
        enc = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = attn.to_q(hidden_states), attn.to_k(enc), attn.to_v(enc)
        hd = c // attn.heads
        q, k, v = [t.view(b, -1, attn.heads, hd).transpose(1, 2) for t in (q, k, v)]
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, c).to(q.dtype)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
        if attn.residual_connection:
            o = o + residual
        return o / attn.rescale_output_factor


def _engine(dev, dtype=torch.float16):
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg = UNetCfg.tiny()
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), random_unet_weights(ocfg, seed=3), dev, dtype=dtype)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 24, 4, generator=g).to(dev).to(dtype)
    enc = torch.randn(2, 77, ocfg.cross_dim, generator=g)
    eng.prepare(torch.tensor([801, 401]), enc.to(dev))
    return eng, x


def test_user_processor_matches_native_kernels(dev):
    eng, x = _engine(dev)
    native = eng.step(x, 0, use_graph=False).float().clone()
    names = list(eng.attn_processors.keys())
    n_blocks = sum(1 for k in eng.w if k.endswith(".attn1.qkv"))
    assert len(names) == 2 * n_blocks and all(n.endswith(".processor") for n in names)
    fam = [n.split(".")[0] for n in names]          # diffusers' registration order: down_blocks, up_blocks, mid_block
    assert fam == sorted(fam, key={"down_blocks": 0, "up_blocks": 1, "mid_block": 2}.get)
    assert sum(".attn1." in n for n in names) == n_blocks and all(v is None for v in eng.attn_processors.values())
    procs = {n: (SdpaProcessor() if (n.startswith("up_blocks") and "attn1" in n) else AttnProcessor()) for n in names}   # Comic_Generation.py:355-370
    eng.set_attn_processor(copy.deepcopy(procs))
    SdpaProcessor.calls = 0
    got = eng.step(x, 0).float()
    n_up = sum(1 for n in names if n.startswith("up_blocks") and "attn1" in n)
    assert SdpaProcessor.calls == n_up > 0
    rel = float((got - native).norm() / native.norm())
    assert rel < 2.6e-3, rel          # measured 2.05e-3 (f16); same products; torch SDPA vs the flash kernel differ in summation order / P rounding
    assert isinstance(eng.attn_processors[[n for n in names if n.startswith("up_blocks") and "attn1" in n][0]], SdpaProcessor)
    # one processor object for every self-attention, then back to the native kernels
    eng.set_attn_processor(SdpaProcessor())
    SdpaProcessor.calls = 0
    got2 = eng.step(x, 0).float()
    assert SdpaProcessor.calls == n_blocks and float((got2 - native).norm() / native.norm()) < 2.6e-3
    eng.set_attn_processor({n: AttnProcessor() for n in names})
    assert eng.self_attn_hook is None
    assert torch.equal(eng.step(x, 0, use_graph=False).float(), native)


def test_registry_errors(dev):
    eng, _ = _engine(dev)
    names = list(eng.attn_processors.keys())
    with pytest.raises(ValueError, match="does not match the number of attention layers"):
        eng.set_attn_processor({names[0]: SdpaProcessor()})
    with pytest.raises(ValueError, match="unknown processor names"):
        eng.set_attn_processor({**{n: None for n in names[:-1]}, "mid_block.attentions.9.transformer_blocks.0.attn1.processor": None})
    with pytest.raises(NotImplementedError):
        eng.set_attn_processor({n: (SdpaProcessor() if ".attn2." in n else None) for n in names})

    class Bad:
        def __call__(self, attn, hidden_states, **kw):
            return hidden_states[:, :1]
    eng.set_attn_processor(Bad())
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 24, 4, generator=g).to(dev).to(torch.float16)
    with pytest.raises(ValueError, match="attention processor at"):
        eng.step(x, 0)
