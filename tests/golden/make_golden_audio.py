"""Generates clap_text_ref.npz and hifigan_ref.npz by running the two `transformers` classes the reference's AudioLDM
pipeline instantiates (spider/models/custom_ad.py:22 imports ClapTextModelWithProjection, SpeechT5HifiGan) on tiny
seeded configs in this container (transformers 5.15.0; the reference pins 4.43.1 / 4.50.0). The fixtures are data:
weights (fp32, bf16-representable), inputs, expected outputs. Run in the build container only:

    python tests/golden/make_golden_audio.py
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))

from oracle.audio import ClapTextCfg, HifiGanCfg, clap_param_shapes, hifigan_param_shapes, random_weights  # noqa: E402


def gen_clap():
    from transformers import ClapTextConfig, ClapTextModelWithProjection
    c = ClapTextCfg.tiny()
    hc = ClapTextConfig(vocab_size=c.vocab, hidden_size=c.hidden, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                        intermediate_size=c.inter, max_position_embeddings=c.max_pos, projection_dim=c.proj_dim,
                        layer_norm_eps=c.eps, pad_token_id=c.pad_id, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = ClapTextModelWithProjection(hc).eval()
    w = random_weights(clap_param_shapes(c), seed=11)
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing), (missing, unexpected)
    ids = torch.tensor([[0, 17, 23, 9, 41, 2, 1, 1, 1, 1], [0, 5, 2, 1, 1, 1, 1, 1, 1, 1], [0, 2, 1, 1, 1, 1, 1, 1, 1, 1],
                        [0, 88, 77, 66, 55, 44, 33, 22, 11, 2]])
    mask = (ids != c.pad_id).long()
    with torch.no_grad():
        out = m(ids, attention_mask=mask).text_embeds
    np.savez_compressed(f"{OUT}/clap_text_ref.npz", ids=ids.numpy(), mask=mask.numpy(), text_embeds=out.numpy(),
                        names=np.array(list(w.keys())), **{f"w{i}": v.numpy() for i, v in enumerate(w.values())})
    print("clap_text_ref.npz", out.shape, float(out.abs().mean()))


def gen_hifigan():
    from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig
    c = HifiGanCfg.tiny()
    hc = SpeechT5HifiGanConfig(model_in_dim=c.model_in_dim, sampling_rate=c.sampling_rate,
                               upsample_initial_channel=c.upsample_initial_channel, upsample_rates=list(c.upsample_rates),
                               upsample_kernel_sizes=list(c.upsample_kernel_sizes),
                               resblock_kernel_sizes=list(c.resblock_kernel_sizes),
                               resblock_dilation_sizes=[list(d) for d in c.resblock_dilation_sizes],
                               leaky_relu_slope=c.leaky_relu_slope, normalize_before=c.normalize_before)
    m = SpeechT5HifiGan(hc).eval()
    w = random_weights(hifigan_param_shapes(c), seed=12)
    m.load_state_dict(w, strict=True)
    mel = torch.randn(2, 13, c.model_in_dim, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        wav = m(mel)
    np.savez_compressed(f"{OUT}/hifigan_ref.npz", mel=mel.numpy(), wav=wav.numpy(), names=np.array(list(w.keys())),
                        **{f"w{i}": v.numpy() for i, v in enumerate(w.values())})
    print("hifigan_ref.npz", wav.shape, float(wav.abs().mean()))


if __name__ == "__main__":
    torch.manual_seed(0)
    gen_clap()
    gen_hifigan()
