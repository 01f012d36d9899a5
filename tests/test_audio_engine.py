"""GPU: the AudioLDM path (SURVEY.md section 8 row a13 + the a10 `class_labels` call site, custom_ad.py:575-581) on the
HIP kernels against the CPU oracles: CLAP text branch and HiFi-GAN vocoder (oracle/audio.py, PINNED to transformers'
own classes through tests/golden/clap_text_ref.npz / hifigan_ref.npz), the class-conditioned UNet form and the mel VAE
(oracle/unet.py, oracle/clip_vae.py -- diffusers restatements, parity unpinned upstream).

Tolerances are relative L2 against the fp32 oracle (bf16 storage, see test_unet_engine.py): 2.5e-2 for one network
evaluation; the vocoder chains ~45 convolutions with a residual stream in bf16, bound 4e-2."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


DT = {"bf16": torch.bfloat16, "f16": torch.float16}       # f16 (+ fp32 residual stream) is what AudioLDMPipeline.from_pretrained loads


def _rel(a, b):
    return float((a.float().cpu() - b).norm() / b.norm())


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return z, {n: torch.from_numpy(z[f"w{i}"]) for i, n in enumerate(z["names"])}


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_clap_text_matches_golden(dev, dtype):
    """engine vs the transformers-generated vectors themselves (and hence vs the oracle, which reproduces them exactly)"""
    from oracle.audio import ClapTextCfg
    from spider_amd.clap import ClapTextConfig, ClapTextEngine
    z, w = _load("clap_text_ref.npz")
    eng = ClapTextEngine(ClapTextConfig(**ClapTextCfg.tiny().__dict__), w, dev, dtype=DT[dtype])
    got = eng.text_embeds(torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]))
    ref = torch.from_numpy(z["text_embeds"])
    assert got.shape == ref.shape
    r = _rel(got, ref)
    print(f"MEASURED clap_text dtype={dtype} rel={r:.5f}")
    bound = {"bf16": 2e-2, "f16": 1e-3}[dtype]          # measured 7.8e-3 / 7.5e-4
    assert r < bound, f"rel L2 {r:.4f}"
    gn = eng.text_embeds(torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]), normalize=True).float().cpu()
    assert torch.allclose(gn.norm(dim=-1), torch.ones(ref.shape[0]), atol=1e-2)
    assert _rel(gn, torch.nn.functional.normalize(ref, dim=-1)) < bound


def test_clap_rejects_left_padding(dev):
    from oracle.audio import ClapTextCfg
    from spider_amd.clap import ClapTextConfig, ClapTextEngine
    z, w = _load("clap_text_ref.npz")
    eng = ClapTextEngine(ClapTextConfig(**ClapTextCfg.tiny().__dict__), w, dev)
    with pytest.raises(ValueError):
        eng.text_embeds(torch.tensor([[1, 0, 5, 2]]), torch.tensor([[0, 1, 1, 1]]))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_hifigan_matches_golden(dev, dtype):
    from oracle.audio import HifiGanCfg
    from spider_amd.vocoder import HifiGanConfig, HifiGanEngine
    z, w = _load("hifigan_ref.npz")
    eng = HifiGanEngine(HifiGanConfig(**HifiGanCfg.tiny().__dict__), w, dev, dtype=DT[dtype])
    got = eng(torch.from_numpy(z["mel"]))
    ref = torch.from_numpy(z["wav"])
    assert got.shape == ref.shape and got.dtype == torch.float32
    r = _rel(got, ref)
    print(f"MEASURED hifigan dtype={dtype} rel={r:.5f}")
    assert r < {"bf16": 4e-2, "f16": 1.5e-3}[dtype], f"rel L2 {r:.4f}"


def test_hifigan_true_shape_runs(dev):
    """full-size vocoder (AudioLDM config, 5 s): output length / range properties"""
    from spider_amd.vocoder import HifiGanConfig, HifiGanEngine
    eng = HifiGanEngine.random_init(HifiGanConfig.audioldm(), dev, seed=0)
    mel = torch.randn(1, 500, 64, generator=torch.Generator().manual_seed(0))
    wav = eng(mel)
    assert wav.shape == (1, 80032) and bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= 1.0
    assert torch.equal(wav, eng(mel)), "deterministic"


@pytest.mark.parametrize("dtype,stream32", [("bf16", False), ("f16", True)])
@pytest.mark.parametrize("hw", [(13, 4), (16, 8)])
def test_audio_unet_step_matches_oracle(dev, hw, dtype, stream32):
    """class-label conditioned UNet, encoder_hidden_states=None; (13,4) exercises the odd-size down/upsampling rule"""
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg = UNetCfg.tiny_audio()
    w = random_unet_weights(ocfg, seed=5)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype], stream32=stream32)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, ocfg.in_ch, *hw, generator=g).bfloat16().float()
    cl = torch.nn.functional.normalize(torch.randn(2, ocfg.class_in, generator=g), dim=-1).bfloat16().float()
    oracle = UNetOracle(ocfg, w)
    ts = torch.tensor([901, 301])
    eng.prepare(ts, None, class_labels=cl.to(dev))
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev).to(DT[dtype])
    for i, t in enumerate(ts):
        ref = oracle.forward(x, t, None, None, cl)
        eager = eng.step(xn, i, use_graph=False).permute(0, 3, 1, 2)
        graph = eng.step(xn, i, use_graph=True).permute(0, 3, 1, 2)
        assert torch.equal(eager.cpu(), graph.cpu())
        r = _rel(eager, ref)
        print(f"MEASURED audio_unet_step dtype={dtype} stream32={stream32} t={int(t)} rel={r:.5f}")
        # bf16 measured 1.15 - 1.27e-2, f16 + fp32 residual stream 1.15 - 1.23e-3 (+20 %)
        assert r < {"bf16": 1.6e-2, "f16": 1.5e-3}[dtype], f"t={int(t)}: rel L2 {r:.4f}"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_audio_vae_decode_matches_oracle(dev, dtype):
    from oracle.clip_vae import VAECfg, random_weights, vae_decode, vae_param_shapes
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    ocfg = VAECfg(latent=8, out_ch=1, block_out=(64, 128, 128), layers_per_block=1, scaling=0.9227)
    w = random_weights(vae_param_shapes(ocfg), seed=7)
    eng = VAEDecoderEngine(VAEConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype])
    lat = torch.randn(1, 8, 13, 4, generator=torch.Generator().manual_seed(8))
    got = eng.decode(lat.to(dev), to_image=False)
    ref = vae_decode(ocfg, w, lat, to_image=False)
    assert got.shape == ref.shape == (1, 1, 52, 16)
    r = _rel(got, ref)
    print(f"MEASURED audio_vae dtype={dtype} rel={r:.5f}")
    assert r < {"bf16": 2.5e-2, "f16": 1.3e-3}[dtype]          # measured 8.1e-3 / 1.03e-3


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_audioldm_pipeline_end_to_end(dev, dtype):
    """prompt -> CLAP -> 6 DDIM steps (CFG) -> mel VAE -> HiFi-GAN -> trimmed waveform, against the oracle chain"""
    from helpers import FakeRobertaTokenizer
    from oracle.audio import (ClapTextCfg, HifiGanCfg, clap_param_shapes, clap_text_embeds, hifigan_forward,
                              hifigan_param_shapes, random_weights)
    from oracle.clip_vae import VAECfg, vae_decode, vae_param_shapes
    from oracle.unet import DDIMOracle, UNetCfg, UNetOracle, denoise_loop, random_unet_weights
    from spider_amd.clap import ClapTextConfig, ClapTextEngine
    from spider_amd.pipelines import AudioLDMPipeline
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    from spider_amd.vocoder import HifiGanConfig, HifiGanEngine
    ccfg = ClapTextCfg(400, 64, 2, 4, 128, 40, 48, 1e-12, 1)
    ucfg = UNetCfg.tiny_audio()                                     # class_in = 48 = CLAP projection width
    vcfg = VAECfg(latent=8, out_ch=1, block_out=(64, 128, 128), layers_per_block=1, scaling=0.9227)
    hcfg = HifiGanCfg(16, 16000, 64, (5, 4, 2), (16, 16, 8), (3, 7), ((1, 3, 5), (1, 3, 5)), 0.1, False)
    wc, wu = random_weights(clap_param_shapes(ccfg), 21), random_unet_weights(ucfg, 22)
    wv, wh = random_weights(vae_param_shapes(vcfg), 23), random_weights(hifigan_param_shapes(hcfg), 24)
    tok = FakeRobertaTokenizer(400)
    d, s32 = DT[dtype], dtype == "f16"
    pipe = AudioLDMPipeline(VAEDecoderEngine(VAEConfig(**vcfg.__dict__), wv, dev, dtype=d), ClapTextEngine(ClapTextConfig(**ccfg.__dict__), wc, dev, dtype=d),
                            tok, UNetEngine(UNetConfig(**ucfg.__dict__), wu, dev, dtype=d, stream32=s32), DDIMScheduler(),
                            HifiGanEngine(HifiGanConfig(**hcfg.__dict__), wh, dev, dtype=d), sample_size=16)
    prompt = ["a dog barking in the rain"]
    secs = 0.12                                                      # hop = 40 samples -> 48 frames -> latent 12 x 4
    lat0 = torch.randn(1, 8, 12, 4, generator=torch.Generator().manual_seed(9))
    out = pipe(prompt=prompt, audio_length_in_s=secs, num_inference_steps=6, guidance_scale=2.5, latents=lat0.clone())
    assert isinstance(out.audios, np.ndarray) and out.audios.shape == (1, int(secs * 16000))
    # oracle chain
    e = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True)
    u = tok([""], padding="max_length", max_length=tok.model_max_length, truncation=True)
    norm = lambda t: torch.nn.functional.normalize(t, dim=-1)
    cl = torch.cat([norm(clap_text_embeds(wc, ccfg, u.input_ids, u.attention_mask)), norm(clap_text_embeds(wc, ccfg, e.input_ids, e.attention_mask))])
    lat = denoise_loop(UNetOracle(ucfg, wu), DDIMOracle(), lat0.clone(), None, 2.5, 6, class_labels=cl)
    mel = vae_decode(vcfg, wv, lat, to_image=False)
    ref = hifigan_forward(wh, hcfg, mel.squeeze(1))[:, :int(secs * 16000)]
    r = _rel(torch.from_numpy(out.audios), ref)
    print(f"MEASURED audioldm_pipeline dtype={dtype} rel={r:.5f}")
    assert r < {"bf16": 8e-2, "f16": 2.2e-3}[dtype], f"waveform rel L2 {r:.4f}"
    # the prompt-embeds entry (spider_decoder.py:150-158) gives the same audio as the text entry
    emb = pipe(prompt, return_prompts_only=True)
    assert emb.shape == (1, 48)
    out2 = pipe(prompt_embeds=emb, audio_length_in_s=secs, num_inference_steps=6, guidance_scale=2.5, latents=lat0.clone())
    assert np.array_equal(out.audios, out2.audios)
    with pytest.raises(ValueError):
        pipe(prompt=prompt, audio_length_in_s=0.001)
    with pytest.raises(ValueError):
        pipe(prompt=prompt, prompt_embeds=emb, audio_length_in_s=secs)
