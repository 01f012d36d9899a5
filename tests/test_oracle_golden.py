"""CPU: pins the oracle (oracle/*.py) against the golden vectors generated from the reference's own code
(tests/golden/make_golden.py) and against transformers' tiny models for the parts the in-tree
modeling_llama.py does not cover (GQA, llama3 rope scaling, qkv bias)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import routing as orouting
from oracle import story as ostory
from oracle.llama import LlamaCfg, LlamaOracle, apply_rope, rmsnorm, rope_table, swiglu_mlp


def _load_llama(golden_dir, seed):
    z = np.load(os.path.join(golden_dir, f"llama_ref_seed{seed}.npz"))
    cfg = LlamaCfg(**json.loads(str(z["cfg"])))
    w = {}
    for i, n in enumerate(z["names"]):
        w[str(n)] = torch.from_numpy(z[f"w{i}"]).view(torch.bfloat16).float()
    return z, cfg, w


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_llama_forward_and_greedy_match_reference(golden_dir, seed):
    z, cfg, w = _load_llama(golden_dir, seed)
    m = LlamaOracle(cfg, w)
    ids = torch.from_numpy(z["ids"])
    pos = torch.arange(ids.shape[1])[None].expand(ids.shape[0], -1)
    logits, _, hid = m.forward(ids, pos, None, None, all_hidden=True)
    assert torch.allclose(logits, torch.from_numpy(z["logits0"]), atol=2e-4, rtol=1e-4)
    ref_h = torch.from_numpy(z["hiddens"])
    for l in range(cfg.layers + 1):
        assert torch.allclose(hid[l], ref_h[l], atol=2e-4, rtol=1e-4), f"hidden state {l}"
    gen, step_logits = m.greedy(ids, 16, return_logits=True)
    assert torch.equal(gen, torch.from_numpy(z["tokens"]))  # bit-exact token ids
    assert torch.allclose(step_logits, torch.from_numpy(z["step_logits"]), atol=5e-4, rtol=1e-4)


def load_llama_d128(golden_dir, k):
    """Reference-generated head_dim-128 fixture k: weights are regenerated from the recorded seed (checksum-guarded)."""
    z = np.load(os.path.join(golden_dir, f"llama_ref_d128_{k}.npz"))
    cfg = LlamaCfg(**json.loads(str(z["cfg"])))
    w = LlamaOracle.random_weights(cfg, seed=int(z["seed"]), std=float(z["std"]))
    wsum = sum(float(v.double().abs().sum()) for v in w.values())
    assert abs(wsum - float(z["wsum"])) <= 1e-9 * float(z["wsum"]), "random_weights no longer reproduces the fixture's weights"
    return z, cfg, w


@pytest.mark.parametrize("k", [0, 1, 2])
def test_llama_d128_forward_and_greedy_match_reference(golden_dir, k):
    """The oracle against the reference's own LlamaForCausalLM at head_dim 128 (the HIP decode kernels' size)."""
    z, cfg, w = load_llama_d128(golden_dir, k)
    assert float(z["margins"].min()) >= 0.1
    m = LlamaOracle(cfg, w)
    ids = torch.from_numpy(z["ids"])
    pos = torch.arange(ids.shape[1])[None].expand(ids.shape[0], -1)
    logits, _, hid = m.forward(ids, pos, None, None, all_hidden=True)
    assert torch.allclose(logits, torch.from_numpy(z["logits0"]), atol=5e-4, rtol=1e-4)
    ref_h = torch.from_numpy(z["hiddens"])
    for l in range(cfg.layers + 1):
        assert torch.allclose(hid[l], ref_h[l], atol=5e-4, rtol=1e-4), f"hidden state {l}"
    gen, step_logits = m.greedy(ids, 16, return_logits=True)
    assert torch.equal(gen, torch.from_numpy(z["tokens"]))
    assert torch.allclose(step_logits, torch.from_numpy(z["step_logits"]), atol=2e-3, rtol=1e-4)


def test_llama_ops_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "llama_ops_ref.npz"))
    t = lambda k: torch.from_numpy(z[k])
    assert torch.allclose(rmsnorm(t("rms_x"), t("rms_w"), float(z["rms_eps"])), t("rms_y"), atol=1e-6)
    cfg = LlamaCfg(head_dim=16, rope_theta=10000.0)
    cs = rope_table(cfg, 64)
    assert torch.allclose(apply_rope(t("rope_q"), cs, t("rope_pos")), t("rope_qe"), atol=1e-5)
    assert torch.allclose(apply_rope(t("rope_k"), cs, t("rope_pos")), t("rope_ke"), atol=1e-5)
    assert torch.allclose(swiglu_mlp(t("mlp_x"), t("mlp_wg"), t("mlp_wu"), t("mlp_wd")), t("mlp_y"), atol=1e-5)


@pytest.mark.parametrize("kind", ["llama3_gqa", "qwen2_bias"])
def test_llama_oracle_matches_transformers_tiny(kind):
    """GQA + llama3 rope scaling / Qwen2 qkv bias live in `transformers` (pinned 4.43.1 / 4.50.0 by the reference's
    requirements files; 5.x here) -- cross-check the restatement against the installed implementation."""
    import transformers
    if kind == "llama3_gqa":
        rs = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                  original_max_position_embeddings=64)
        cfg = LlamaCfg(64, 2, 4, 2, 16, 128, 101, 500000.0, rs, 1e-5, False, 256)
        try:
            hf = transformers.LlamaConfig(vocab_size=101, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                          num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5,
                                          max_position_embeddings=256, rope_theta=500000.0, rope_scaling=dict(rs),
                                          attention_bias=False, tie_word_embeddings=False)
        except Exception as e:  # pragma: no cover
            pytest.skip(f"transformers config API changed: {e}")
        model = transformers.LlamaForCausalLM(hf)
    else:
        cfg = LlamaCfg(64, 2, 4, 2, 16, 128, 101, 1000000.0, None, 1e-6, True, 256)
        hf = transformers.Qwen2Config(vocab_size=101, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                      num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-6,
                                      max_position_embeddings=256, rope_theta=1000000.0, tie_word_embeddings=False)
        model = transformers.Qwen2ForCausalLM(hf)
    model = model.float().eval()
    w = LlamaOracle.random_weights(cfg, seed=3, std=0.3)
    missing, unexpected = model.load_state_dict(w, strict=False)
    assert not unexpected and not [k for k in missing if "rotary" not in k], (missing, unexpected)
    # confirm the installed model really uses the rope parameters we think it does
    ids = torch.randint(3, 101, (2, 90), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = model(input_ids=ids).logits
    m = LlamaOracle(cfg, w)
    pos = torch.arange(90)[None].expand(2, -1)
    got, _, _ = m.forward(ids, pos, None, None)
    assert torch.allclose(got, ref, atol=3e-4, rtol=1e-4), float((got - ref).abs().max())
    with torch.no_grad():
        gen_ref = model.generate(ids, max_new_tokens=8, do_sample=False, num_beams=1, use_cache=True,
                                 pad_token_id=0, eos_token_id=None)[:, 90:]
    assert torch.equal(m.greedy(ids, 8), gen_ref)


def test_left_padded_greedy_equals_unpadded():
    cfg = LlamaCfg(64, 2, 4, 2, 16, 128, 101, 10000.0, None, 1e-6, True, 256)
    m = LlamaOracle(cfg, LlamaOracle.random_weights(cfg, seed=5, std=0.3))
    ids = torch.randint(3, 101, (1, 9), generator=torch.Generator().manual_seed(2))
    a = m.greedy(ids, 6)
    padded = torch.cat([torch.zeros(1, 4, dtype=torch.long), ids], 1)
    am = torch.cat([torch.zeros(1, 4, dtype=torch.long), torch.ones(1, 9, dtype=torch.long)], 1)
    b = m.greedy(padded, 6, attn_mask=am)
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------- routing
def test_routing_matches_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "routing_ref.json")))
    assert len(ref["cases"]) >= 60
    for c in ref["cases"]:
        text = c["text"]
        assert orouting.get_llm_text_modality(text, ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX"]) == c["modality"]
        for m_, r in c["res"].items():
            assert orouting.get_llm_text_res(text, m_) == r
        answers, ptext, calls, _ = orouting.route(text)
        assert answers == c["answers"]
        assert ptext == c["predictions_text"]
        assert [list(x) for x in calls] == c["calls"]
    for s in ref["story"]:
        gp, pa, sn = orouting.extract_story_elements(s["text"])
        assert (gp, pa, sn) == (s["general_prompt"], s["prompt_array"], s["style_name"]), s["text"]


def test_oracle_routing_matches_reference_on_random_grammar_strings(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "routing_ref_fuzz.json")))
    for c in ref["cases"]:
        text = c["text"]
        assert orouting.get_llm_text_modality(text, ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX"]) == c["modality"], repr(text)
        for m_, r in c["res"].items():
            assert orouting.get_llm_text_res(text, m_) == r, repr(text)
        if not c["none_mode"]:
            answers, ptext, calls, _ = orouting.route(text)
            assert answers == c["answers"] and ptext == c["predictions_text"] and [list(x) for x in calls] == c["calls"], repr(text)
    for s in ref["story"]:
        assert orouting.extract_story_elements(s["text"]) == (s["general_prompt"], s["prompt_array"], s["style_name"]), repr(s["text"])


def test_routing_known_answers():
    # spider_decoder_infer.py:139-142 and spider_decoder.py:284-295
    answers, ptext, _, _ = orouting.route("<IMAGE>apple</IMAGE><VIDEO>dog</VIDEO><AUDIO>cat</AUDIO>")
    assert answers == ["<IMAGE>apple</IMAGE><VIDEO>dog</VIDEO><AUDIO>cat</AUDIO>"]
    assert ptext == {'IMAGE': ['apple'], 'VIDEO': ['dog'], 'AUDIO': ['cat'], 'MASK': [], 'BOX': [], 'IMAGESTORY': [],
                     'IMAGESTORY_prompts': []}
    assert orouting.get_llm_text_res("<MASK>apple</MASK>", "MASK") == ["apple"]
    assert orouting.get_llm_text_modality("<IMAGE>a</IMAGE><VIDEO>b</VIDEO><AUDIO>c</AUDIO>",
                                          ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX"]) == ["IMAGE", "VIDEO", "AUDIO"]


# ------------------------------------------------------------------------------------------- StoryDiffusion
def _aw(z, prefix, heads):
    t = lambda k: torch.from_numpy(z[f"{prefix}_{k}"])
    return ostory.AttnWeights(t("to_q.weight"), t("to_k.weight"), t("to_v.weight"), t("to_out.0.weight"),
                              t("to_out.0.bias"), heads)


def test_story_masks_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "story_ref.npz"))
    for i in range(3):
        h, w = [int(v) for v in z[f"mask{i}_hw"]]
        m1, m4 = ostory.cal_attn_mask_xl(5, 4, 0.5, 0.5, h, w, torch.from_numpy(z[f"mask{i}_rand1024"]),
                                         torch.from_numpy(z[f"mask{i}_rand4096"]))
        assert torch.equal(m1, torch.from_numpy(z[f"mask{i}_m1024"]))
        assert torch.equal(m4, torch.from_numpy(z[f"mask{i}_m4096"]))
    # worked 5x5 example of gradio_utils.py:241-295: keep = [1,0,1,1,0], id rows limited to first 4 groups
    u = torch.tensor([0.4, 0.9, 0.1, 0.2, 0.8])
    m1, _ = ostory.cal_attn_mask_xl(5, 4, 0.5, 0.5, 32, 32, u, torch.rand(20))
    exp = torch.tensor([[1, 0, 1, 1, 0], [1, 1, 1, 1, 0], [1, 0, 1, 1, 0], [1, 0, 1, 1, 0], [1, 0, 1, 1, 1]]).bool()
    assert torch.equal(m1, exp)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_story_processor_calls_match_reference(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "story_ref.npz"))
    N, C, heads, hh, ww = [int(v) for v in z[f"proc{tag}_cfg"]]
    aw = _aw(z, f"proc{tag}", heads)
    hs = torch.from_numpy(z[f"proc{tag}_hs"])
    mask = torch.from_numpy(z[f"proc{tag}_mask"])
    assert torch.allclose(ostory.call1(aw, hs, None, mask), torch.from_numpy(z[f"proc{tag}_y1"]), atol=2e-5)
    assert torch.allclose(ostory.call2(aw, hs, None, None), torch.from_numpy(z[f"proc{tag}_y2"]), atol=2e-5)


def test_story_write_sequence_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "story_ref.npz"))
    C, heads, hh, ww = [int(v) for v in z["seq_cfg"]]
    aw_a, aw_b = _aw(z, "seq_sa", heads), _aw(z, "seq_sb", heads)
    coins = [c for row in z["seq_coins"] for c in row if c >= 0]
    it = iter(coins)
    st = ostory.StoryState(total_count=2, height=hh, width=ww, coin=lambda: float(next(it)))
    # masks are injected per step from the recorded keep rows (row 0 of the reference masks == keep | own block 0)
    step_masks = []
    n1, n4 = (hh // 32) * (ww // 32), (hh // 16) * (ww // 16)
    pa, pb = ostory.ProcessorOracle(), ostory.ProcessorOracle()

    def set_masks(step):
        k1 = torch.from_numpy(z["seq_keep1024"][step]).clone(); k4 = torch.from_numpy(z["seq_keep4096"][step]).clone()
        # row 0 has its own block forced True; rebuild the full masks from the other rows' rule
        # (a column in block 0 was kept iff row 1's entry is True)
        def rebuild(k, n):
            b = k[None].repeat(5, 1)
            return b
        return k1, k4

    outs_a, outs_b = [], []
    for step in range(7):
        # rebuild full masks exactly: use u = 0 where kept else 1 for blocks 1..3; block 0 needs row 1 -> not recorded,
        # so compare only through the processor outputs with masks reconstructed from the keep rows of rows != own.
        k1, k4 = set_masks(step)
        # reference mask row r (block i): keep | own(i). Row 0 gives keep for blocks 1..4 exactly; for block 0 use
        # the recorded row (own block forced True) -- the write-phase slice [:4N,:4N] of row-blocks 1..3 needs the
        # true keep of block 0, which equals what any row of block 1 shows. It is stored in the fixture generator as
        # row 0 only, so regenerate through torch's RNG instead (same container image -> same stream).
        pass
    # Regenerate the reference stream deterministically instead (CPU torch.rand after manual_seed(2047)):
    torch.manual_seed(2047)
    st.uniforms = lambda n: torch.rand((1, n), dtype=torch.float32).reshape(-1)
    st.regen_masks()
    g = None
    for step in range(7):
        assert torch.equal(st.mask1024[0], torch.from_numpy(z["seq_keep1024"][step]))
        assert torch.equal(st.mask4096[0], torch.from_numpy(z["seq_keep4096"][step]))
        ya = pa(st, aw_a, torch.from_numpy(z["seq_xa"][step]))
        yb = pb(st, aw_b, torch.from_numpy(z["seq_xb"][step]))
        assert torch.allclose(ya, torch.from_numpy(z["seq_ya"][step]), atol=2e-5), step
        assert torch.allclose(yb, torch.from_numpy(z["seq_yb"][step]), atol=2e-5), step
    assert st.cur_step == 7


def test_clip_oracle_matches_transformers_tiny():
    """The CLIP text encoder lives in `transformers` (reference call site custom_sd.py:306-310); pin the restatement
    against the installed CLIPTextModel on a tiny random config."""
    import transformers
    from oracle.clip_vae import CLIPCfg, clip_param_shapes, clip_text_forward, random_weights
    c = CLIPCfg.tiny()
    hf = transformers.CLIPTextConfig(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.inter,
                                     num_hidden_layers=c.layers, num_attention_heads=c.heads,
                                     max_position_embeddings=c.max_pos, hidden_act="quick_gelu", layer_norm_eps=c.eps,
                                     pad_token_id=1, bos_token_id=0, eos_token_id=2)
    model = transformers.CLIPTextModel(hf).float().eval()
    w = random_weights(clip_param_shapes(c), seed=4)
    keys = set(model.state_dict().keys())
    wl = w if "text_model.final_layer_norm.weight" in keys else {k.replace("text_model.", "", 1): v for k, v in w.items()}
    missing, unexpected = model.load_state_dict(wl, strict=False)   # transformers 5.x dropped the text_model. prefix
    assert not unexpected and not [k for k in missing if "position_ids" not in k], (missing, unexpected)
    ids = torch.randint(3, c.vocab, (2, 77), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = model(input_ids=ids).last_hidden_state
    got = clip_text_forward(c, w, ids)
    assert torch.allclose(got, ref, atol=2e-4, rtol=1e-4), float((got - ref).abs().max())


# ------------------------------------------------------------------------------------------ AudioLDM side networks
def _load_audio(name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    return z, {n: torch.from_numpy(z[f"w{i}"]) for i, n in enumerate(z["names"])}


def test_clap_text_oracle_matches_transformers_vectors():
    """oracle/audio.py vs ClapTextModelWithProjection outputs (tests/golden/make_golden_audio.py), incl. padded rows"""
    from oracle.audio import ClapTextCfg, clap_position_ids, clap_text_embeds
    z, w = _load_audio("clap_text_ref.npz")
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    got = clap_text_embeds(w, ClapTextCfg.tiny(), ids, mask)
    assert torch.allclose(got, torch.from_numpy(z["text_embeds"]), atol=1e-5, rtol=1e-5)
    # a padded row equals the same prompt encoded without padding (what the HIP engine relies on)
    n = int(mask[1].sum())
    alone = clap_text_embeds(w, ClapTextCfg.tiny(), ids[1:2, :n], mask[1:2, :n])
    assert torch.allclose(alone, got[1:2], atol=1e-5)
    assert clap_position_ids(torch.tensor([[0, 5, 2, 1, 1]]), 1).tolist() == [[2, 3, 4, 1, 1]]


def test_hifigan_oracle_matches_transformers_vectors():
    from oracle.audio import HifiGanCfg, hifigan_forward
    z, w = _load_audio("hifigan_ref.npz")
    got = hifigan_forward(w, HifiGanCfg.tiny(), torch.from_numpy(z["mel"]))
    ref = torch.from_numpy(z["wav"])
    assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-5, rtol=1e-5)
    # length rule of the transposed convs: L -> (L-1)*r - 2*((k-r)//2) + k per stage
    L = z["mel"].shape[1]
    for r, k in zip((5, 4, 2), (16, 16, 8)):
        L = (L - 1) * r - 2 * ((k - r) // 2) + k
    assert got.shape[1] == L


def test_audio_unet_oracle_shapes_and_upsample_rule():
    """class-label conditioned UNet form on an odd latent (AudioLDM's 125-row case in miniature)"""
    import torch.nn.functional as F
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights, unet_param_shapes
    cfg = UNetCfg.tiny_audio()
    S = unet_param_shapes(cfg)
    assert S["class_embedding.weight"] == (cfg.temb_dim, cfg.class_in)
    assert S["down_blocks.0.resnets.0.time_emb_proj.weight"][1] == 2 * cfg.temb_dim          # class_embeddings_concat
    assert S["down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (128, 128)   # cross dim = block width
    u = UNetOracle(cfg, random_unet_weights(cfg, 0))
    g = torch.Generator().manual_seed(0)
    x, cl = torch.randn(2, 8, 13, 4, generator=g), torch.randn(2, cfg.class_in, generator=g)
    y = u.forward(x, torch.tensor(500), None, None, cl)
    assert y.shape == x.shape and bool(torch.isfinite(y).all())
    y2 = u.forward(x, torch.tensor(500), None, None, cl * 0.5)
    assert not torch.allclose(y, y2), "class labels must condition the output"
    a = torch.randn(1, 3, 32, 5, generator=g)   # nearest to 2n-1 == crop of the exact 2x (what the fused conv addressing assumes)
    assert torch.equal(F.interpolate(a, size=(63, 9), mode="nearest"), F.interpolate(a, scale_factor=2.0, mode="nearest")[:, :, :63, :9])


def test_unet3d_oracle_structure_and_loop_reshapes():
    """UNet3D restatement: parameter count of the published checkpoint shape, temporal mixing, and the equivalence the HIP
    path relies on -- keeping latents as [B*F,C,h,w] for the whole loop == the reference's per-step reshapes
    (custom_vd.py:684-692)."""
    import math
    from oracle.unet import DDIMOracle
    from oracle.unet3d import UNet3DCfg, UNet3DOracle, random_unet3d_weights, tensor2vid, unet3d_param_shapes, video_denoise_loop
    n = sum(math.prod(s) for s in unet3d_param_shapes(UNet3DCfg.zeroscope()).values())
    assert abs(n / 1e9 - 1.411) < 0.01, n      # the "1.7b" text-to-video checkpoint: 1.41 B UNet parameters
    cfg = UNet3DCfg.tiny()
    u = UNet3DOracle(cfg, random_unet3d_weights(cfg, 0))
    g = torch.Generator().manual_seed(0)
    x, enc = torch.randn(2, 4, 3, 8, 12, generator=g), torch.randn(2, 77, cfg.cross_dim, generator=g)
    y = u.forward(x, torch.tensor(500), enc)
    assert y.shape == x.shape and bool(torch.isfinite(y).all())
    x2 = x.clone(); x2[:, :, 2] += 1.0                      # perturb the LAST frame only
    y2 = u.forward(x2, torch.tensor(500), enc)
    assert not torch.allclose(y[:, :, 0], y2[:, :, 0], atol=1e-4), "temporal layers must mix information across frames"
    # frames-as-batch loop == reshaping loop
    lat0 = torch.randn(1, 4, 3, 8, 12, generator=g)
    enc2 = torch.randn(2, 77, cfg.cross_dim, generator=g)
    ref = video_denoise_loop(u, DDIMOracle(), lat0.clone(), enc2, 9.0, 2)
    sched = DDIMOracle(); ts = sched.set_timesteps(2)
    lat = lat0.permute(0, 2, 1, 3, 4).reshape(3, 4, 8, 12) * sched.init_noise_sigma
    for t in ts:
        l5 = lat.view(1, 3, 4, 8, 12).permute(0, 2, 1, 3, 4)
        e = u.forward(torch.cat([l5] * 2), t, enc2)
        eu, ec = e.chunk(2)
        eps = (eu + 9.0 * (ec - eu)).permute(0, 2, 1, 3, 4).reshape(3, 4, 8, 12)
        lat = sched.step(eps, t, lat)
    assert torch.allclose(lat.view(1, 3, 4, 8, 12).permute(0, 2, 1, 3, 4), ref, atol=1e-5)
    # tensor2vid: batch tiled horizontally, uint8
    v = torch.zeros(2, 3, 2, 4, 5); v[1] = 1.0
    fr = tensor2vid(v)
    assert len(fr) == 2 and fr[0].shape == (4, 10, 3) and fr[0][:, :5].max() == 127 and fr[0][:, 5:].min() == 255


# ------------------------------------------------------------------------------------------ trained-Spider output side (N3)
def _moe_golden():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "moe_proj_ref.npz"))
    return z, json.loads(str(z["mods"])), json.loads(str(z["capture"]))


def test_moe_projector_oracle_matches_reference():
    """oracle/moe_proj.py vs the reference's own TextFcLayerMoE (executed by tests/golden/make_golden.py::gen_moe)"""
    from oracle.moe_proj import moe_forward, moe_param_shapes, random_moe_weights
    z, mods, _ = _moe_golden()
    w = random_moe_weights(int(z["in_dim"]), mods, int(z["seed"]))
    assert sum(v.numel() for v in w.values()) > 80e6 and set(w) == set(moe_param_shapes(int(z["in_dim"]), mods))
    for t in "abc":
        y = moe_forward(w, torch.from_numpy(z[f"{t}_x"]), str(z[f"{t}_mod"]))
        assert torch.allclose(y, torch.from_numpy(z[f"{t}_y"]), atol=2e-5, rtol=1e-4), t


def test_qformer_textfc_oracle_matches_reference():
    """oracle/moe_proj.py:qformer_textfc_forward vs the reference's own Q-Former blocks (BertEmbeddings / BertLayer of
    spider/models/Qformer.py, executed by tests/golden/make_golden.py::gen_qformer between TextFcLayer's fc and model layers)"""
    from oracle.moe_proj import qformer_param_shapes, qformer_textfc_forward, random_qformer_weights
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "textfc_qformer_ref.npz"))
    dims = [int(z[k]) for k in ("in_dim", "out_dim", "n_query")]
    w = random_qformer_weights(*dims, inter=int(z["inter"]), seed=int(z["seed"]))
    assert set(w) == set(qformer_param_shapes(*dims, inter=int(z["inter"])))
    assert abs(float(sum(v.double().abs().sum() for v in w.values())) - float(z["w_checksum"])) < 1e-6 * float(z["w_checksum"])
    for t in "abc":
        y = qformer_textfc_forward(w, torch.from_numpy(z[f"{t}_x"]))
        ref = torch.from_numpy(z[f"{t}_y"])
        assert y.shape == ref.shape == (z[f"{t}_x"].shape[0], dims[2], dims[1])
        assert torch.allclose(y, ref, atol=2e-5, rtol=1e-4), t


class _FakeLLM:
    device = torch.device("cpu")

    def embed_tokens(self, ids):
        return ids.float().unsqueeze(-1).expand(*ids.shape, 8) * 0.5


class _SignalTok:
    pad_token_id, bos_token_id = 0, 1

    def __init__(self, begin, end):
        self.begin, self.end = begin, end

    def __call__(self, text, return_tensors="pt", add_special_tokens=False):
        mod = text.strip("</>")
        class R:
            input_ids = torch.tensor([[self.end[mod] if text.startswith("</") else self.begin[mod]]])
        return R()


def test_signal_token_capture_matches_reference():
    """product TrainedSpider.preparing_output_embeds_infer and the oracle's span arithmetic vs the reference's function run on
    a stub model (spider.py:1413-1463)"""
    from oracle.moe_proj import capture_spans
    from spider_amd.spider_trained import TrainedSpider
    _, _, cap = _moe_golden()
    seq, H, L = cap["seq"], cap["H"], cap["L"]
    hs = tuple(tuple(torch.full((1, 1, H), float(step * 10 + layer)) for layer in range(L)) for step in range(len(seq)))
    class Outputs:
        sequences = torch.tensor([seq])
        hidden_states = hs
    ts = TrainedSpider(_FakeLLM(), _SignalTok(cap["begin"], cap["end"]), [], cap["alignment_layer"], cap["modality_tokens"])
    for case in cap["cases"]:
        mod, mi = case["modality"], case["modality_i"]
        m, h, i_, ht, it = ts.preparing_output_embeds_infer({"TaskPrompt": [f"[{mod}]"]}, Outputs(), modality=mod, modality_i=mi)
        assert m == mod
        for got, ref in ((h, case["hidden"]), (i_, case["inputs"]), (ht, case["hidden_text"]), (it, case["inputs_text"])):
            assert len(got) == len(ref)
            for g, r in zip(got, ref):
                assert torch.equal(g.float(), torch.tensor(r)), (mod, mi)
        (lo, hi), (tlo, thi) = capture_spans(seq[1:], cap["begin"][mod], cap["end"][mod], cap["modality_tokens"][mod], mi)
        assert [v[0] for v in case["hidden"][0][0]] == [float(s * 10 + (L - 1)) for s in range(lo, hi)]
        assert [v[0] for v in case["hidden_text"][0][0]] == [float(s * 10 + (L - 1)) for s in range(tlo, thi)]
    assert TrainedSpider.split_placeholder("<IMAGE><IMAGE-Placeholder></IMAGE> a dog") == ["<IMAGE>", "<IMAGE-Placeholder>", "</IMAGE> a dog"]


def test_mrope_oracle_matches_transformers_qwen25_omni_text_model():
    """multimodal RoPE restatement vs transformers' Qwen2_5OmniThinkerTextModel (tiny random, 3-row position ids)"""
    from transformers.models.qwen2_5_omni import modeling_qwen2_5_omni as m
    hc = m.Qwen2_5OmniTextConfig(vocab_size=200, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                                 num_key_value_heads=1, max_position_embeddings=512, rope_theta=1000000.0, rms_norm_eps=1e-6,
                                 rope_scaling={"mrope_section": [16, 24, 24], "rope_type": "default", "type": "default"})
    torch.manual_seed(0)
    model = m.Qwen2_5OmniThinkerTextModel(hc).eval()
    ocfg = LlamaCfg(256, 2, 2, 1, 128, 512, 200, 1000000.0, None, 1e-6, True, 512, False, (16, 24, 24))
    w = {("model." + k): v.detach().float() for k, v in model.state_dict().items()}
    w["lm_head.weight"] = w["model.embed_tokens.weight"]
    x = torch.randn(2, 9, 256)
    pos = torch.stack([torch.tensor([0, 1, 2, 2, 2, 2, 6, 7, 8]), torch.tensor([0, 1, 2, 2, 3, 3, 6, 7, 8]),
                       torch.tensor([0, 1, 2, 3, 2, 3, 6, 7, 8])])[:, None, :].expand(3, 2, 9).contiguous()
    with torch.no_grad():
        ref = model(inputs_embeds=x, position_ids=pos).last_hidden_state
    _, _, hs = LlamaOracle(ocfg, w).forward(None, pos, None, None, inputs_embeds=x, all_hidden=True)
    assert torch.allclose(hs[-1], ref, atol=2e-5, rtol=1e-4)
    _, _, hs1 = LlamaOracle(ocfg, w).forward(None, pos[0], None, None, inputs_embeds=x, all_hidden=True)
    assert not torch.allclose(hs1[-1], ref, atol=1e-4), "1-D positions must differ from the 3-component ones"


# ---------------------------------------------------------------------------------------------- Qwen2.5-Omni input towers (N4)
def _load_towers():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen_towers_ref.npz"))
    vw = {str(n): torch.from_numpy(z[f"vw{i}"]) for i, n in enumerate(z["v_names"])}
    aw = {str(n): torch.from_numpy(z[f"aw{i}"]) for i, n in enumerate(z["a_names"])}
    return z, vw, aw


def test_vision_tower_oracle_matches_transformers_vectors():
    """oracle/qwen_towers.py:vision_forward against transformers' own Qwen2_5OmniVisionEncoder (fixture made by
    tests/golden/make_golden_towers.py): ragged windows, a 2-frame clip and an exactly divisible grid in one packed call."""
    from oracle.qwen_towers import VisionCfg, vision_forward
    z, vw, _ = _load_towers()
    last, pooled = vision_forward(VisionCfg.tiny(), vw, torch.from_numpy(z["v_pixel_values"]), z["v_grid"].tolist())
    assert torch.allclose(last, torch.from_numpy(z["v_last_hidden"]), atol=2e-5, rtol=1e-5)
    assert torch.allclose(pooled, torch.from_numpy(z["v_pooler"]), atol=2e-5, rtol=1e-5)


def test_audio_tower_oracle_matches_transformers_vectors():
    """oracle/qwen_towers.py:audio_forward against transformers' Qwen2_5OmniAudioEncoder: three audios (2 full chunks +
    tail, exactly one chunk, odd post-CNN length whose last frame the stride-2 pooling drops)."""
    from oracle.qwen_towers import AudioCfg, audio_forward, audio_output_lengths
    z, _, aw = _load_towers()
    got = audio_forward(AudioCfg.tiny(), aw, torch.from_numpy(z["a_features"]), z["a_lens"].tolist())
    assert got.shape[0] == sum(audio_output_lengths(z["a_lens"].tolist()))
    assert torch.allclose(got, torch.from_numpy(z["a_out"]), atol=2e-5, rtol=1e-5)


def test_rope_index_oracle_matches_transformers_vectors():
    """get_rope_index restatement, bit-exact (integer work): image, audio+image, video, left-padded batch, and a video with
    its audio track interleaved (use_audio_in_video, the flag qwen2.5omni_spider_web.py:468 passes)."""
    import importlib.util
    from oracle.qwen_towers import OmniTokenIds, get_rope_index
    spec = importlib.util.spec_from_file_location("mgt", os.path.join(os.path.dirname(__file__), "golden", "make_golden_towers.py"))
    mgt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgt)
    z, _, _ = _load_towers()
    for i, cse in enumerate(mgt.rope_cases()):
        ids = torch.tensor(cse["ids"])
        mask = torch.tensor(cse["mask"]) if "mask" in cse else torch.ones_like(ids)
        pos, delta = get_rope_index(OmniTokenIds(), 2, ids, cse["img"], cse["vid"], mask, cse["av"], cse["aud"], cse["spg"])
        assert np.array_equal(pos.numpy(), z[f"r{i}_pos"]), f"case {i}"
        assert np.array_equal(delta.numpy(), z[f"r{i}_delta"]), f"case {i}"
    # text-only prompts take the cumulative-mask branch
    am = torch.tensor([[0, 0, 1, 1, 1]])
    pos, delta = get_rope_index(OmniTokenIds(), 2, torch.tensor([[0, 0, 5, 6, 7]]), None, None, am)
    assert pos[:, 0].tolist() == [[1, 1, 0, 1, 2]] * 3 and int(delta) == 0
