"""Multimodal input side of the Qwen2.5-Omni thinker: vision tower, audio tower, rope index, splice.

CPU tests: the product's host index logic against the vectors transformers produced (tests/golden/qwen_towers_ref.npz) and
against the oracle. GPU tests: the HIP engines against the fp32 oracle (itself pinned to those vectors) on the golden
inputs and on larger seeded cases; integer work is exact, bf16 activations within the stated relative L2."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import qwen_towers as oq

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    z = np.load(os.path.join(HERE, "golden", "qwen_towers_ref.npz"))
    vw = {str(n): torch.from_numpy(z[f"vw{i}"]) for i, n in enumerate(z["v_names"])}
    aw = {str(n): torch.from_numpy(z[f"aw{i}"]) for i, n in enumerate(z["a_names"])}
    return z, vw, aw


def _rope_cases():
    spec = importlib.util.spec_from_file_location("mgt", os.path.join(HERE, "golden", "make_golden_towers.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.rope_cases()


# ---------------------------------------------------------------------------------------------- host logic (CPU)
def test_product_rope_index_matches_transformers_vectors():
    from spider_amd.qwen_omni import OmniTokenIds, get_rope_index
    z, _, _ = _golden()
    for i, cse in enumerate(_rope_cases()):
        ids = torch.tensor(cse["ids"])
        mask = torch.tensor(cse["mask"]) if "mask" in cse else torch.ones_like(ids)
        pos, delta = get_rope_index(OmniTokenIds(), 2, ids, cse["img"], cse["vid"], mask, cse["av"], cse["aud"], cse["spg"])
        assert np.array_equal(pos.numpy(), z[f"r{i}_pos"]), f"case {i}"
        assert np.array_equal(delta.numpy(), z[f"r{i}_delta"]), f"case {i}"


@pytest.mark.parametrize("grid", [[(1, 6, 10)], [(2, 8, 4), (1, 4, 4)], [(1, 36, 52)], [(3, 14, 18), (1, 2, 2)]])
def test_product_vision_index_logic_matches_oracle(grid):
    from spider_amd import qwen_omni as q
    for merge, window, patch in ((2, 16, 4), (2, 112, 14)):
        wi, cu = q.vision_window_index(grid, merge, window, patch)
        wo, co = oq.vision_window_index(grid, merge, window, patch)
        assert torch.equal(wi, wo) and cu == co
        assert sorted(wi.tolist()) == list(range(len(wi))) and cu[-1] == sum(t * h * w for t, h, w in grid)
        assert torch.equal(q.vision_position_ids(grid, merge), oq.vision_position_ids(grid, merge))


def test_product_audio_length_logic():
    from spider_amd import qwen_omni as q
    for lens in ([47, 20, 33], [200], [201, 3], [1000, 999]):
        assert q.audio_chunk_lengths(lens, 100) == oq.audio_chunk_lengths(lens, 100)
        assert sum(q.audio_chunk_lengths(lens, 100)) == sum(lens)
        assert q.audio_output_lengths(lens) == oq.audio_output_lengths(lens)
    assert q.audio_chunk_lengths([200, 250], 100) == [200, 200, 50]


def test_varlen_tiles_cover_every_segment_once():
    from spider_amd import ops
    t = ops.varlen_tiles([0, 5, 5 + 300, 5 + 300 + 128], "cpu").tolist()
    assert t == [[0, 5, 0, 5], [5, 128, 5, 300], [133, 128, 5, 300], [261, 44, 5, 300], [305, 128, 305, 128]]
    with pytest.raises(ValueError):
        ops.varlen_tiles([0, 4, 2], "cpu")


# ---------------------------------------------------------------------------------------------- HIP engines (GPU)
def _rel(a, b):
    return float((a.float().cpu() - b).norm() / b.norm())


@pytest.mark.gpu
def test_varlen_attention_and_rope_rows_match_torch(dev):
    """Packed variable-length attention (segments of 5 / 300 / 128 / 64 rows, d = 80 as in the 7B vision tower) and the per-row
    half-rotation RoPE against plain fp32 torch."""
    from spider_amd import ops
    g = torch.Generator().manual_seed(0)
    cu = [0, 5, 305, 433, 497]
    T, nh, d = cu[-1], 3, 80
    qkv = (torch.randn(T, 3 * nh * d, generator=g)).bfloat16()
    ang = torch.randn(T, d // 2, generator=g)
    cs = torch.cat([ang.cos(), ang.sin()], 1).contiguous()
    qf, kf, vf = [t.float().reshape(T, nh, d) for t in qkv.split(nh * d, 1)]
    cos, sin = ang.cos().repeat(1, 2)[:, None], ang.sin().repeat(1, 2)[:, None]
    qr = (qf * cos + oq._rot_half(qf) * sin).bfloat16().float()
    kr = (kf * cos + oq._rot_half(kf) * sin).bfloat16().float()
    ref = oq._seg_attention(qr, kr, vf, cu, d ** -0.5).reshape(T, nh * d)
    x = qkv.to(dev)
    ops.rope_rows_(x[:, :nh * d], cs.to(dev), nh)
    ops.rope_rows_(x[:, nh * d:2 * nh * d], cs.to(dev), nh)
    assert torch.equal(x[:, :nh * d].cpu().float().reshape(T, nh, d), qr)          # fp32 rotate, one bf16 rounding: exact
    assert torch.equal(x[:, 2 * nh * d:].cpu(), qkv[:, 2 * nh * d:])              # v untouched
    out = ops.attention_varlen(x[:, :nh * d], x[:, nh * d:2 * nh * d], x[:, 2 * nh * d:], nh, ops.varlen_tiles(cu, dev))
    assert _rel(out, ref) < 1e-2


@pytest.mark.gpu
def test_vision_tower_matches_oracle_on_golden_inputs(dev):
    from spider_amd.qwen_omni import VisionTowerConfig, VisionTowerEngine
    z, vw, _ = _golden()
    oc = oq.VisionCfg.tiny()
    eng = VisionTowerEngine(VisionTowerConfig(**oc.__dict__), vw, dev)
    last, pooled = eng(torch.from_numpy(z["v_pixel_values"]), z["v_grid"].tolist(), return_last_hidden=True)
    assert tuple(pooled.shape) == tuple(z["v_pooler"].shape)
    assert _rel(last, torch.from_numpy(z["v_last_hidden"])) < 2e-2
    assert _rel(pooled, torch.from_numpy(z["v_pooler"])) < 2e-2
    with pytest.raises(ValueError):
        eng(torch.from_numpy(z["v_pixel_values"])[:-1], z["v_grid"].tolist())
    with pytest.raises(ValueError):
        eng(torch.from_numpy(z["v_pixel_values"]), [[1, 5, 10]])


@pytest.mark.gpu
def test_vision_tower_mid_size_with_padded_intermediate(dev):
    """head_dim 80 and an MLP width that is not a multiple of 8 (the 7B tower's 3420 -> padded with zero rows), 448 x 644 px
    image + a 2-frame clip: windows of 64 patches, ragged edge windows, full attention over 1472-patch frames."""
    from spider_amd.qwen_omni import VisionTowerConfig, VisionTowerEngine
    oc = oq.VisionCfg(depth=4, hidden=160, heads=2, inter=212, in_channels=3, patch=14, temporal_patch=2, merge=2, window=112,
                      out_hidden=192, fullatt=(1, 3), eps=1e-6)
    w = oq.random_weights(oq.vision_param_shapes(oc), seed=31)
    grid = [[1, 32, 46], [2, 6, 10]]
    n = sum(t * h * ww for t, h, ww in grid)
    px = torch.randn(n, oc.patch_dim, generator=torch.Generator().manual_seed(1)).bfloat16().float()
    _, ref = oq.vision_forward(oc, w, px, grid)
    got = VisionTowerEngine(VisionTowerConfig(**oc.__dict__), w, dev)(px, grid)
    assert _rel(got, ref) < 2e-2


@pytest.mark.gpu
def test_audio_tower_matches_oracle_on_golden_inputs(dev):
    from spider_amd.qwen_omni import AudioTowerConfig, AudioTowerEngine
    z, _, aw = _golden()
    oc = oq.AudioCfg.tiny()
    eng = AudioTowerEngine(AudioTowerConfig(**oc.__dict__), aw, dev)
    got = eng(torch.from_numpy(z["a_features"]), z["a_lens"].tolist())
    assert tuple(got.shape) == tuple(z["a_out"].shape)
    assert _rel(got, torch.from_numpy(z["a_out"])) < 2e-2
    with pytest.raises(ValueError):
        eng(torch.from_numpy(z["a_features"]), [47, 20, 34])


@pytest.mark.gpu
def test_audio_tower_mid_size_full_chunk_geometry(dev):
    """The 7B tower's chunk geometry (n_window 100 -> 200-frame chunks of 100 tokens, head_dim 64, 128 mel bins) with two
    audios: 3 full chunks + 37-frame tail, and one exact chunk."""
    from spider_amd.qwen_omni import AudioTowerConfig, AudioTowerEngine
    oc = oq.AudioCfg(mel=128, layers=3, heads=4, ffn=512, d_model=256, max_pos=1500, n_window=100, out_dim=320)
    w = oq.random_weights(oq.audio_param_shapes(oc), seed=32)
    lens = [637, 200]
    feats = torch.randn(oc.mel, sum(lens), generator=torch.Generator().manual_seed(2)).bfloat16().float()
    ref = oq.audio_forward(oc, w, feats, lens)
    got = AudioTowerEngine(AudioTowerConfig(**oc.__dict__), w, dev)(feats, lens)
    assert got.shape[0] == sum(oq.audio_output_lengths(lens)) == 159 + 50
    assert _rel(got, ref) < 2e-2


@pytest.mark.gpu
def test_thinker_image_audio_text_prompt_matches_oracle(dev):
    """End to end: image + audio + text prompt -> towers -> splice -> (t, h, w) positions -> prefill + greedy decode, against
    the oracle chain (oracle towers -> oracle splice -> oracle rope index -> LlamaOracle with mrope)."""
    from oracle.llama import LlamaCfg, LlamaOracle
    from spider_amd.llm import LlamaEngine, LLMConfig
    from spider_amd.qwen_omni import AudioTowerConfig, AudioTowerEngine, OmniTokenIds, QwenOmniThinker, VisionTowerConfig, VisionTowerEngine
    H = 256
    lcfg = LlamaCfg(H, 2, 4, 2, 128, 512, 400, 1000000.0, None, 1e-6, True, 512, False, (16, 24, 24))
    lw = LlamaOracle.random_weights(lcfg, seed=7, std=0.08)
    vc = oq.VisionCfg(depth=2, hidden=64, heads=2, inter=88, in_channels=3, patch=4, temporal_patch=2, merge=2, window=16,
                      out_hidden=H, fullatt=(1,), eps=1e-6)
    ac = oq.AudioCfg(mel=16, layers=2, heads=2, ffn=96, d_model=32, max_pos=40, n_window=10, out_dim=H)
    vw = oq.random_weights(oq.vision_param_shapes(vc), seed=8)
    aw = oq.random_weights(oq.audio_param_shapes(ac), seed=9)
    tok = OmniTokenIds(image=390, video=391, audio=392, vision_start=393, audio_start=394)
    otok = oq.OmniTokenIds(image=390, video=391, audio=392, vision_start=393, audio_start=394)
    grid, alen = [[1, 4, 6]], [47]
    n_img, n_aud = 6, oq.audio_output_lengths(alen)[0]
    ids = torch.tensor([[5, 6, 394] + [392] * n_aud + [395, 7, 393] + [390] * n_img + [396, 8, 9, 10]])
    g = torch.Generator().manual_seed(3)
    px = torch.randn(24, vc.patch_dim, generator=g).bfloat16().float()
    feats = torch.randn(ac.mel, alen[0], generator=g).bfloat16().float()
    # oracle chain
    oracle = LlamaOracle(lcfg, lw)
    emb = lw["model.embed_tokens.weight"][ids]
    emb = oq.splice_features(otok, ids, emb, audio_features=oq.audio_forward(ac, aw, feats, alen).bfloat16().float(),
                             image_embeds=oq.vision_forward(vc, vw, px, grid)[1].bfloat16().float())
    pos, _ = oq.get_rope_index(otok, 2, ids, grid, None, torch.ones_like(ids), False, alen, None)
    logits, kv, hs = oracle.forward(None, pos, None, None, inputs_embeds=emb, all_hidden=True)
    ref_tok = [int(logits[0, -1].argmax())]
    p = int(pos.max()) + 1
    for n in range(5):
        lg, kv, _ = oracle.forward(torch.tensor([[ref_tok[-1]]]), torch.tensor([[p + n]]), kv, None)
        ref_tok.append(int(lg[0, -1].argmax()))
    # HIP chain
    llm = LlamaEngine(LLMConfig(**lcfg.__dict__), lw, dev, max_batch=1, max_len=128)
    thinker = QwenOmniThinker(llm, VisionTowerEngine(VisionTowerConfig(**vc.__dict__), vw, dev),
                              AudioTowerEngine(AudioTowerConfig(**ac.__dict__), aw, dev), tok)
    e2, p2 = thinker.prepare_inputs(ids, torch.ones_like(ids), pixel_values=px, image_grid_thw=torch.tensor(grid),
                                    input_features=feats[None], feature_attention_mask=torch.ones(1, alen[0], dtype=torch.long))
    assert torch.equal(p2, pos)
    assert _rel(e2, emb) < 2e-2
    text_rows = (ids[0] != 390) & (ids[0] != 392)
    assert torch.equal(e2[0, text_rows].float().cpu(), emb[0, text_rows])          # token rows are untouched by the splice
    out = thinker.generate(ids, torch.ones_like(ids), max_new_tokens=6, pixel_values=px, image_grid_thw=torch.tensor(grid),
                           input_features=feats[None], feature_attention_mask=torch.ones(1, alen[0], dtype=torch.long),
                           spk="Chelsie", return_dict_in_generate=True, return_logits=True)
    assert torch.equal(out.sequences[:, :ids.shape[1]].cpu(), ids)
    gen = out.sequences[0, ids.shape[1]:].tolist()
    first_div = next((i for i, (a, b) in enumerate(zip(gen, ref_tok)) if a != b), None)
    if first_div is not None:   # only a near-tie of the logits may flip a greedy token
        top2 = out.logits[0, first_div].float().cpu().topk(2).values
        assert float(top2[0] - top2[1]) < 0.08, (gen, ref_tok)
    with pytest.raises(ValueError):   # placeholder count must equal the tower's row count
        thinker.prepare_inputs(ids[:, :-8], None, pixel_values=px, image_grid_thw=torch.tensor(grid))


@pytest.mark.gpu
def test_from_pretrained_reads_the_omni_checkpoint_layout(dev, tmp_path):
    """config.json with the nested thinker_config (text / vision / audio) + safetensors with the `thinker.*` prefixes, plus
    talker weights that must be ignored: the three engines load through from_pretrained and reproduce engines built directly."""
    import json
    from safetensors.torch import save_file
    from oracle.llama import LlamaCfg, LlamaOracle
    from spider_amd.llm import LlamaEngine, LLMConfig
    from spider_amd.qwen_omni import AudioTowerConfig, AudioTowerEngine, QwenOmniThinker, VisionTowerConfig, VisionTowerEngine
    lcfg = LlamaCfg(256, 2, 4, 2, 128, 512, 400, 1000000.0, None, 1e-6, True, 512, False, (16, 24, 24))
    lw = LlamaOracle.random_weights(lcfg, seed=7, std=0.08)
    vc, ac = oq.VisionCfg.tiny(), oq.AudioCfg.tiny()
    vc.out_hidden = ac.out_dim = 256
    vw = oq.random_weights(oq.vision_param_shapes(vc), seed=8)
    aw = oq.random_weights(oq.audio_param_shapes(ac), seed=9)
    sd = {("thinker." + k): v.contiguous() for k, v in lw.items()}
    sd.update({"thinker.visual." + k: v.contiguous() for k, v in vw.items()})
    sd.update({"thinker.audio_tower." + k: v.contiguous() for k, v in aw.items()})
    sd["talker.model.embed_tokens.weight"] = torch.zeros(4, 4)
    save_file({k: v.to(torch.bfloat16) for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    cfg = {"model_type": "qwen2_5_omni", "thinker_config": {
        "text_config": {"model_type": "qwen2_5_omni_text", "hidden_size": 256, "num_hidden_layers": 2, "num_attention_heads": 4,
                        "num_key_value_heads": 2, "head_dim": 128, "intermediate_size": 512, "vocab_size": 400, "rope_theta": 1000000.0,
                        "rope_scaling": {"mrope_section": [16, 24, 24], "rope_type": "default"}, "rms_norm_eps": 1e-6,
                        "max_position_embeddings": 512, "tie_word_embeddings": False},
        "vision_config": {"depth": vc.depth, "hidden_size": vc.hidden, "num_heads": vc.heads, "intermediate_size": vc.inter,
                          "in_channels": 3, "patch_size": vc.patch, "temporal_patch_size": 2, "spatial_merge_size": 2,
                          "window_size": vc.window, "out_hidden_size": 256, "fullatt_block_indexes": list(vc.fullatt)},
        "audio_config": {"num_mel_bins": ac.mel, "encoder_layers": ac.layers, "encoder_attention_heads": ac.heads,
                         "encoder_ffn_dim": ac.ffn, "d_model": ac.d_model, "max_source_positions": ac.max_pos,
                         "n_window": ac.n_window, "output_dim": 256},
        "image_token_index": 390, "video_token_index": 391, "audio_token_index": 392, "vision_start_token_id": 393,
        "audio_start_token_id": 394, "position_id_per_seconds": 25, "seconds_per_chunk": 2}}
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    th = QwenOmniThinker.from_pretrained(str(tmp_path), dev, max_len=128)
    assert th.ids.image == 390 and th.ids.audio_start == 394 and th.llm.cfg.mrope_section == (16, 24, 24) and th.llm.cfg.qkv_bias
    assert th.vision.cfg.fullatt == tuple(vc.fullatt) and th.audio.cfg.n_window == ac.n_window
    g = torch.Generator().manual_seed(4)
    px = torch.randn(24, vc.patch_dim, generator=g).bfloat16().float()
    feats = torch.randn(ac.mel, 47, generator=g).bfloat16().float()
    v_ref = VisionTowerEngine(VisionTowerConfig(**vc.__dict__), vw, dev)(px, [[1, 4, 6]])
    a_ref = AudioTowerEngine(AudioTowerConfig(**ac.__dict__), aw, dev)(feats, [47])
    assert torch.equal(th.vision(px, [[1, 4, 6]]), v_ref) and torch.equal(th.audio(feats, [47]), a_ref)
    ids = torch.tensor([[5, 6, 393] + [390] * 6 + [396, 8, 9]])
    ref = QwenOmniThinker(LlamaEngine(LLMConfig(**lcfg.__dict__), lw, dev, max_batch=1, max_len=128), th.vision, th.audio, th.ids)
    kw = dict(pixel_values=px, image_grid_thw=torch.tensor([[1, 4, 6]]), max_new_tokens=4)
    assert torch.equal(th.generate(ids, None, **kw), ref.generate(ids, None, **kw))


@pytest.mark.gpu
def test_thinker_left_padded_batch_equals_single_prompts(dev):
    """`padding=True` batches of the processor (qwen2.5omni_spider_web.py:466): a left-padded batch of an image prompt and a
    shorter text-only prompt generates exactly what each prompt generates on its own (pad slots are masked, their dummy
    rotary positions never enter a visible key, decode positions continue from each row's own maximum)."""
    from oracle.llama import LlamaCfg, LlamaOracle
    from spider_amd.llm import LlamaEngine, LLMConfig
    from spider_amd.qwen_omni import OmniTokenIds, QwenOmniThinker, VisionTowerConfig, VisionTowerEngine
    H = 256
    lcfg = LlamaCfg(H, 2, 4, 2, 128, 512, 400, 1000000.0, None, 1e-6, True, 512, False, (16, 24, 24))
    lw = LlamaOracle.random_weights(lcfg, seed=17, std=0.08)
    vc = oq.VisionCfg(depth=2, hidden=64, heads=2, inter=88, in_channels=3, patch=4, temporal_patch=2, merge=2, window=16,
                      out_hidden=H, fullatt=(1,), eps=1e-6)
    vw = oq.random_weights(oq.vision_param_shapes(vc), seed=18)
    tok = OmniTokenIds(image=390, video=391, audio=392, vision_start=393, audio_start=394)
    llm = LlamaEngine(LLMConfig(**lcfg.__dict__), lw, dev, max_batch=2, max_len=128)
    th = QwenOmniThinker(llm, VisionTowerEngine(VisionTowerConfig(**vc.__dict__), vw, dev), None, tok)
    px = torch.randn(24, vc.patch_dim, generator=torch.Generator().manual_seed(5)).bfloat16().float()
    grid = torch.tensor([[1, 4, 6]])
    a = [5, 6, 393] + [390] * 6 + [396, 8, 9, 10, 11, 12, 13, 14, 15]      # 18 tokens, one image
    b = [21, 22, 23, 24, 25, 26, 27, 28, 29, 30]                            # 10 tokens, text only
    pad = len(a) - len(b)
    ids = torch.tensor([a, [0] * pad + b])
    am = torch.tensor([[1] * len(a), [0] * pad + [1] * len(b)])
    both = th.generate(ids, am, max_new_tokens=6, pixel_values=px, image_grid_thw=grid)
    one_a = th.generate(torch.tensor([a]), torch.ones(1, len(a), dtype=torch.long), max_new_tokens=6, pixel_values=px, image_grid_thw=grid)
    one_b = th.generate(torch.tensor([b]), torch.ones(1, len(b), dtype=torch.long), max_new_tokens=6)
    assert torch.equal(both[0], one_a[0])
    assert torch.equal(both[1, pad:], one_b[0])


@pytest.mark.gpu
def test_spider_free_infer_with_an_image_in_the_request(dev):
    """configs[3] ("image + text in"): the `predict` flow (qwen2.5omni_spider_web.py:458-521) through spider_amd.SpiderFreeInfer with
    a request that carries an image -- processor output with pixel_values / image_grid_thw -> vision tower -> mRoPE positions -> generate ->
    batch_decode -> Decoders-Controller; the request's image reaches `ask_info['Image_ori_array']` (:494-499) and pipelined requests of
    the same geometry give the serial result."""
    import numpy as np
    from oracle.llama import LlamaCfg, LlamaOracle
    from spider_amd import SpiderDecoderInfer, SpiderFreeInfer
    from spider_amd.llm import LlamaEngine, LLMConfig
    from spider_amd.qwen_omni import OmniTokenIds, QwenOmniThinker, VisionTowerConfig, VisionTowerEngine
    from benchkit.synthetic import SyntheticOmniProcessor
    H = 256
    lcfg = LlamaCfg(H, 2, 4, 2, 128, 512, 400, 1000000.0, None, 1e-6, True, 512, False, (16, 24, 24))
    lw = LlamaOracle.random_weights(lcfg, seed=27, std=0.08)
    vc = oq.VisionCfg(depth=2, hidden=64, heads=2, inter=88, in_channels=3, patch=4, temporal_patch=2, merge=2, window=16,
                      out_hidden=H, fullatt=(1,), eps=1e-6)
    vw = oq.random_weights(oq.vision_param_shapes(vc), seed=28)
    tok = OmniTokenIds(image=390, video=391, audio=392, vision_start=393, audio_start=394)
    th = QwenOmniThinker(LlamaEngine(LLMConfig(**lcfg.__dict__), lw, dev, max_batch=1, max_len=128),
                         VisionTowerEngine(VisionTowerConfig(**vc.__dict__), vw, dev), None, tok)
    px = torch.randn(24, vc.patch_dim, generator=torch.Generator().manual_seed(6)).bfloat16().float()
    image = np.arange(8 * 12 * 3, dtype=np.uint8).reshape(8, 12, 3)

    class Proc(SyntheticOmniProcessor):          # what Qwen2_5OmniProcessor emits for one image: placeholder tokens + patches + grid
        def __call__(self, text=None, audios=None, images=None, videos=None, return_tensors="pt", padding=True):
            ids = torch.tensor([[5, 6, 393] + [390] * 6 + [396, 8, 9]])
            self.prompt_len = ids.shape[1]
            out = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
            if images is not None:
                out.update(pixel_values=px, image_grid_thw=torch.tensor([[1, 4, 6]]))
            return out

    seen = []

    class Pipe:                                   # image decoder stand-in: records the captions it is asked for
        def __call__(self, prompt=None, **kw):
            seen.append(list(prompt))
            class O:
                images = [f"img:{p}" for p in prompt]
            return O()

    dinf = SpiderDecoderInfer({"model": dict(type="spider_decoder", pipelines={"IMAGE": Pipe()}, device=str(dev))})
    asks = []
    gen = dinf.spider_decoder.generate
    dinf.spider_decoder.generate = lambda samples, *a: (asks.append(dict(samples)), gen(samples, *a))[1]
    infer = SpiderFreeInfer(th, Proc(vocab=400), dinf, device=dev, generate_kwargs=dict(max_new_tokens=6, eos_token_id=[], spk="Chelsie",
                                                                                       use_audio_in_video=True),
                            process_mm_info=lambda messages, use_audio_in_video: (None, [image], None))
    msgs = [{"role": "user", "content": [{"type": "image", "image": "x.png"}, {"type": "text", "text": "what is this?"}]}]
    res = infer(msgs)
    direct = th.generate(torch.tensor([[5, 6, 393] + [390] * 6 + [396, 8, 9]]), None, max_new_tokens=6, eos_token_id=[], pixel_values=px,
                         image_grid_thw=torch.tensor([[1, 4, 6]]))
    assert torch.equal(res.text_ids, direct[0].cpu()), "the class generates exactly what the thinker generates for the processor output"
    assert np.array_equal(asks[0]["Image_ori_array"][0], image) and asks[0]["llm_text_all"] == [res.response]
    assert res.predictions["IMAGE"] == [f"img:{res.predictions_text['IMAGE'][0]}"] and seen[-1] == res.predictions_text["IMAGE"]
    got = list(infer.pipelined([msgs, msgs, msgs]))
    assert [r.response for r in got] == [res.response] * 3 and all(torch.equal(r.text_ids, res.text_ids) for r in got)
