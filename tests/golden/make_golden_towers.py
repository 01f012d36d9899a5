"""Generates qwen_towers_ref.npz by running the `transformers` classes behind the reference's
`Qwen2_5OmniModel.generate(**inputs)` call with images / audios (qwen2.5omni_spider_web.py:461-468) on tiny seeded
configs in this container (transformers 5.15.0; the reference pins 4.50.0): Qwen2_5OmniVisionEncoder,
Qwen2_5OmniAudioEncoder and Qwen2_5OmniPreTrainedModelForConditionalGeneration.get_rope_index. The fixture is data:
weights (fp32, bf16-representable), inputs, expected outputs. Run in the build container only:

    python tests/golden/make_golden_towers.py
"""
import functools
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))

from oracle.qwen_towers import (AudioCfg, OmniTokenIds, VisionCfg, audio_param_shapes, random_weights,  # noqa: E402
                                vision_param_shapes)

VISION_GRIDS = [[1, 6, 10], [2, 8, 4], [1, 4, 4]]       # ragged windows, a 2-frame clip, an exactly divisible grid
AUDIO_LENS = [47, 20, 33]                               # 2 full chunks + tail; exactly one chunk; odd post-CNN length


def gen_vision(store):
    from transformers.models.qwen2_5_omni.configuration_qwen2_5_omni import Qwen2_5OmniVisionEncoderConfig
    from transformers.models.qwen2_5_omni.modeling_qwen2_5_omni import Qwen2_5OmniVisionEncoder
    c = VisionCfg.tiny()
    hc = Qwen2_5OmniVisionEncoderConfig(depth=c.depth, hidden_size=c.hidden, hidden_act="silu", intermediate_size=c.inter,
                                        num_heads=c.heads, in_channels=c.in_channels, patch_size=c.patch,
                                        spatial_merge_size=c.merge, temporal_patch_size=c.temporal_patch, window_size=c.window,
                                        out_hidden_size=c.out_hidden, fullatt_block_indexes=list(c.fullatt))
    hc._attn_implementation = "eager"
    m = Qwen2_5OmniVisionEncoder(hc).eval().float()
    w = random_weights(vision_param_shapes(c), seed=21)
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected and not [k for k in missing if "inv_freq" not in k], (missing, unexpected)
    grid = torch.tensor(VISION_GRIDS)
    n = int(grid.prod(-1).sum())
    px = torch.randn(n, c.patch_dim, generator=torch.Generator().manual_seed(5)).bfloat16().float()
    with torch.no_grad():
        o = m(px, grid_thw=grid)
    store.update(v_pixel_values=px.numpy(), v_grid=grid.numpy(), v_last_hidden=o.last_hidden_state.numpy(),
                 v_pooler=o.pooler_output.numpy(), v_names=np.array(list(w.keys())),
                 **{f"vw{i}": v.numpy() for i, v in enumerate(w.values())})
    print("vision", tuple(o.pooler_output.shape), float(o.pooler_output.abs().mean()))


def gen_audio(store):
    from transformers.models.qwen2_5_omni.configuration_qwen2_5_omni import Qwen2_5OmniAudioEncoderConfig
    from transformers.models.qwen2_5_omni.modeling_qwen2_5_omni import Qwen2_5OmniAudioEncoder
    c = AudioCfg.tiny()
    hc = Qwen2_5OmniAudioEncoderConfig(num_mel_bins=c.mel, encoder_layers=c.layers, encoder_attention_heads=c.heads,
                                       encoder_ffn_dim=c.ffn, d_model=c.d_model, activation_function="gelu",
                                       max_source_positions=c.max_pos, n_window=c.n_window, output_dim=c.out_dim)
    hc._attn_implementation = "eager"
    m = Qwen2_5OmniAudioEncoder(hc).eval().float()
    w = random_weights(audio_param_shapes(c), seed=22)
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected and not [k for k in missing if "positional_embedding" not in k], (missing, unexpected)
    lens = torch.tensor(AUDIO_LENS)
    feats = torch.randn(c.mel, int(lens.sum()), generator=torch.Generator().manual_seed(6)).bfloat16().float()
    after = (lens - 1) // 2 + 1
    with torch.no_grad():
        o = m(feats, feature_lens=lens, aftercnn_lens=after).last_hidden_state
    store.update(a_features=feats.numpy(), a_lens=lens.numpy(), a_out=o.numpy(), a_names=np.array(list(w.keys())),
                 **{f"aw{i}": v.numpy() for i, v in enumerate(w.values())})
    print("audio", tuple(o.shape), float(o.abs().mean()))


def rope_cases():
    t = OmniTokenIds()
    V, A, I, VS, AS = t.video, t.audio, t.image, t.vision_start, t.audio_start
    VE, AE = 151653, 151648            # vision_end / audio_end: ordinary tokens as far as get_rope_index is concerned
    txt = lambda n, s=100: list(range(s, s + n))
    cases = []
    # 1: text, image (2x(4x6) patches -> 6 tokens), text
    cases.append(dict(ids=[txt(3) + [VS] + [I] * 6 + [VE] + txt(4, 200)], img=[[1, 4, 6]], vid=None, aud=None, av=False, spg=None))
    # 2: audio then image; audio feature length 47 -> 12 tokens
    cases.append(dict(ids=[txt(2) + [AS] + [A] * 12 + [AE] + txt(1, 300) + [VS] + [I] * 4 + [VE] + txt(5, 400)],
                      img=[[1, 4, 4]], vid=None, aud=[47], av=False, spg=None))
    # 3: video without audio, 3 temporal grids, second_per_grid 2.0
    cases.append(dict(ids=[txt(1) + [VS] + [V] * 12 + [VE] + txt(2, 500)], img=None, vid=[[3, 4, 4]], aud=None, av=False, spg=[2.0]))
    # 4: batch of two rows with left padding (attention_mask), image in each
    r1 = txt(2) + [VS] + [I] * 2 + [VE] + txt(3, 600)
    r2 = [0, 0] + [VS] + [I] * 2 + [VE] + txt(3, 700)
    cases.append(dict(ids=[r1, r2], img=[[1, 2, 4], [1, 4, 2]], vid=None, aud=None, av=False, spg=None, mask=[[1] * 9, [0, 0] + [1] * 7]))
    # 5: video with its audio track interleaved (use_audio_in_video): 4 temporal grids x (2x2 merged) = 16 video tokens,
    #    audio length 120 -> 30 tokens; second_per_grid 1.0 -> 25 position ids per grid, chunks of 50
    cases.append(dict(ids=[txt(2) + [VS, AS] + [V] * 16 + [A] * 30 + [AE, VE] + txt(2, 800)], img=None, vid=[[4, 4, 4]], aud=[120],
                      av=True, spg=[1.0]))
    return cases


def gen_rope(store):
    from transformers.models.qwen2_5_omni.modeling_qwen2_5_omni import Qwen2_5OmniPreTrainedModelForConditionalGeneration as Cls
    t = OmniTokenIds()
    cfg = types.SimpleNamespace(image_token_id=t.image, video_token_id=t.video, audio_token_id=t.audio,
                                vision_start_token_id=t.vision_start, audio_start_token_id=t.audio_start,
                                position_id_per_seconds=t.position_id_per_seconds, seconds_per_chunk=t.seconds_per_chunk)
    fake = types.SimpleNamespace(spatial_merge_size=2, config=cfg)
    fake.get_llm_pos_ids_for_vision = functools.partial(Cls.get_llm_pos_ids_for_vision, fake)
    fake.get_chunked_index = functools.partial(Cls.get_chunked_index, fake)
    for i, cse in enumerate(rope_cases()):
        ids = torch.tensor(cse["ids"])
        mask = torch.tensor(cse["mask"]) if "mask" in cse else torch.ones_like(ids)
        pos, delta = Cls.get_rope_index(
            fake, ids, torch.tensor(cse["img"]) if cse["img"] else None, torch.tensor(cse["vid"]) if cse["vid"] else None,
            mask, cse["av"], torch.tensor(cse["aud"]) if cse["aud"] else None,
            torch.tensor(cse["spg"]) if cse["spg"] else None)
        store[f"r{i}_pos"] = pos.numpy()
        store[f"r{i}_delta"] = delta.numpy()
        print("rope case", i, tuple(pos.shape), int(pos.max()))


if __name__ == "__main__":
    store = {}
    gen_vision(store)
    gen_audio(store)
    gen_rope(store)
    np.savez_compressed(f"{OUT}/qwen_towers_ref.npz", **store)
    print("wrote", f"{OUT}/qwen_towers_ref.npz", os.path.getsize(f"{OUT}/qwen_towers_ref.npz"), "bytes")
