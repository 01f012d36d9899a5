"""GPU: StableDiffusionPipeline call contract (custom_sd.py:476-667) on tiny engines with a stand-in tokenizer, and the
SpiderDecoder image path end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pipe(dev, dtype=torch.float16, stream32=True, precise=0):
    """default: the mode StableDiffusionPipeline.from_pretrained loads (f16 operands, fp32 residual stream in the UNet)"""
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
    from spider_amd.pipelines import StableDiffusionPipeline
    from spider_amd.unet import UNetConfig, UNetEngine
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    from helpers import FakeTokenizer
    uc, cc, vc = UNetCfg.tiny(), CLIPCfg.tiny(), VAECfg.tiny()
    return StableDiffusionPipeline(UNetEngine(UNetConfig(**uc.__dict__), random_unet_weights(uc, seed=1), dev, dtype=dtype, stream32=stream32, precise=precise),
                                   VAEDecoderEngine(VAEConfig(**vc.__dict__), random_weights(vae_param_shapes(vc), seed=2), dev, dtype=dtype),
                                   CLIPTextEngine(CLIPTextConfig(**cc.__dict__), random_weights(clip_param_shapes(cc), seed=3), dev, dtype=dtype),
                                   FakeTokenizer(), sample_size=16), (uc, cc, vc)


def test_pipeline_contract(dev):
    pipe, (uc, cc, vc) = _pipe(dev)
    g = torch.Generator(device=dev).manual_seed(7)
    out = pipe(prompt=["an apple on a table"], guidance_scale=7.5, num_inference_steps=6, generator=g)
    assert len(out.images) == 1 and out.images[0].size == (64, 64)   # tiny VAE: 3 levels -> x4
    # return_prompts_only: encoder states WITHOUT the CFG concat (custom_sd.py:589-605)
    emb = pipe(["an apple on a table"], return_prompts_only=True)
    assert emb.shape == (1, 77, cc.hidden)
    # prompt_embeds path == prompt path (same latents)
    lat = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(1))
    a = pipe(prompt=["an apple on a table"], num_inference_steps=5, latents=lat, output_type="np").images
    b = pipe(prompt_embeds=emb, num_inference_steps=5, latents=lat, output_type="np").images
    assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        pipe(prompt=["x"], height=100, width=64)
    # left truncation of over-long prompts (custom_sd.py:267-276): keeps the LAST model_max_length tokens
    long_prompt = " ".join(f"word{i}" for i in range(120))
    ids = pipe._tokenize([long_prompt])
    assert ids.shape == (1, 77)


@pytest.mark.parametrize("dtype,stream32,precise", [("bf16", False, 0), ("f16", True, 0), ("f16", True, 1), ("f16", True, 2)])
def test_pipeline_matches_oracle_end_to_end(dev, dtype, stream32, precise):
    """text -> CLIP -> PNDM loop -> VAE, HIP engines vs the fp32 oracle pieces chained the same way (precise = 1 / 2: the UNet's precise
    modes through the pipeline's own denoising loop, fp32 latents in)."""
    from oracle.clip_vae import clip_text_forward, vae_decode
    from oracle.unet import PNDMOracle, UNetOracle, denoise_loop
    pipe, (uc, cc, vc) = _pipe(dev, {"bf16": torch.bfloat16, "f16": torch.float16}[dtype], stream32, precise)
    prompt = ["a cozy cabin in the snow"]
    ids_c = pipe._tokenize(prompt)
    ids_u = pipe.tokenizer([""], padding="max_length", max_length=77, truncation=True).input_ids
    wu = {k: v.float().cpu() for k, v in pipe.unet.w.items()}  # not used: oracle takes the original fp32 dicts below
    from oracle.clip_vae import CLIPCfg, VAECfg, clip_param_shapes, random_weights, vae_param_shapes
    from oracle.unet import UNetCfg, random_unet_weights
    enc = torch.cat([clip_text_forward(cc, random_weights(clip_param_shapes(cc), seed=3), ids_u),
                     clip_text_forward(cc, random_weights(clip_param_shapes(cc), seed=3), ids_c)])
    lat0 = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    lat = denoise_loop(UNetOracle(uc, random_unet_weights(uc, seed=1)), PNDMOracle(), lat0, enc, 7.5, 6)
    ref = vae_decode(vc, random_weights(vae_param_shapes(vc), seed=2), lat).permute(0, 2, 3, 1).numpy()
    got = pipe(prompt=prompt, guidance_scale=7.5, num_inference_steps=6, latents=lat0, output_type="np").images
    err = np.abs(got - ref)
    print(f"MEASURED sd_pipeline dtype={dtype} stream32={stream32} precise={precise} mean_abs={float(err.mean()):.5f}")
    assert err.mean() < {"bf16": 2e-2, "f16": 6e-4}[dtype] and got.shape == ref.shape, float(err.mean())


def test_spider_decoder_image_path(dev):
    from spider_amd import routing
    from spider_amd.spider_decoder import SpiderDecoder
    pipe, _ = _pipe(dev)
    dec = SpiderDecoder(diffusion_modules={}, pipelines=dict(IMAGE=pipe))
    a, p, pt = dec.generate({"llm_text_all": ["Sure: <IMAGE>a red car</IMAGE>"]}, *routing.new_outputs())
    assert pt["IMAGE"] == ["a red car"] and len(p["IMAGE"]) == 1 and p["IMAGE"][0].size == (64, 64)
    dec2 = SpiderDecoder(diffusion_modules={}, pipelines=dict(IMAGE=pipe), get_prompt_embed_for_diffusion=True)
    a, p2, _ = dec2.generate({"llm_text_all": ["<IMAGE>a red car</IMAGE>"]}, *routing.new_outputs())
    assert len(p2["IMAGE"]) == 1


def test_spider_decoder_audio_and_video_paths(dev):
    """routing -> decode_audio / decode_video with the reference's defaults (spider_decoder.py:122,145): 40 steps,
    5.0 s of audio, 16 frames; tiny engines, real HIP pipelines (registry names 'ad' / 'vd')."""
    from helpers import tiny_audio_pipe, tiny_video_pipe
    from spider_amd import routing
    from spider_amd.registry import registry
    from spider_amd.pipelines import AudioLDMPipeline, TextToVideoSDPipeline
    from spider_amd.spider_decoder import SpiderDecoder
    assert registry.get_model_class("ad") is AudioLDMPipeline and registry.get_model_class("vd") is TextToVideoSDPipeline
    dec = SpiderDecoder(diffusion_modules={}, pipelines=dict(AUDIO=tiny_audio_pipe(dev), VIDEO=tiny_video_pipe(dev)))
    txt = "Here you go. <AUDIO>rain on a tin roof</AUDIO> and <VIDEO>waves at sunset</VIDEO>"
    a, p, pt = dec.generate({"llm_text_all": [txt]}, *routing.new_outputs())
    assert pt["AUDIO"] == ["rain on a tin roof"] and pt["VIDEO"] == ["waves at sunset"]
    assert len(p["AUDIO"]) == 1 and p["AUDIO"][0].shape == (80000,) and np.isfinite(p["AUDIO"][0]).all()
    assert len(p["VIDEO"]) == 1 and len(p["VIDEO"][0]) == 16 and p["VIDEO"][0][0].shape == (320, 576, 3)
    # prompt-embeds control path (spider_decoder.py:126-135, :149-158)
    dec2 = SpiderDecoder(diffusion_modules={}, pipelines=dict(AUDIO=tiny_audio_pipe(dev)), get_prompt_embed_for_diffusion=True)
    _, p2, _ = dec2.generate({"llm_text_all": ["<AUDIO>rain on a tin roof</AUDIO>"]}, *routing.new_outputs())
    assert p2["AUDIO"][0].shape == (80000,)


def test_spider_decoder_generate_batch_runs_each_decoder_once(dev):
    """SpiderDecoder.generate_batch on the real HIP pipelines (tiny engines): three responses, each with IMAGE + AUDIO + VIDEO
    captions, decoded by ONE pipeline call per modality (CFG batch 2 x 3); every sample gets its own containers with entries shaped
    exactly like the one-sample `generate` call's."""
    from helpers import tiny_audio_pipe, tiny_video_pipe
    from spider_amd.spider_decoder import SpiderDecoder
    pipe, _ = _pipe(dev)
    pipes = dict(IMAGE=pipe, AUDIO=tiny_audio_pipe(dev), VIDEO=tiny_video_pipe(dev))
    calls = {m: 0 for m in pipes}
    for m, p in pipes.items():
        orig = p.__class__.__call__
        def counted(self, *a, _m=m, _o=orig, **k):
            calls[_m] += 1
            return _o(self, *a, **k)
        p.__class__ = type(p.__class__.__name__ + "Counted", (p.__class__,), {"__call__": counted})
    dec = SpiderDecoder(diffusion_modules={}, pipelines=pipes)
    texts = [f"ok <IMAGE>car {i}</IMAGE> <AUDIO>rain {i}</AUDIO> <VIDEO>sea {i}</VIDEO>" for i in range(3)]
    outs = dec.generate_batch([{"llm_text_all": [t]} for t in texts])
    assert calls == dict(IMAGE=1, AUDIO=1, VIDEO=1)
    assert len(outs) == 3
    for i, (a, p, pt) in enumerate(outs):
        assert a == [texts[i]] and pt["IMAGE"] == [f"car {i}"] and pt["AUDIO"] == [f"rain {i}"] and pt["VIDEO"] == [f"sea {i}"]
        assert len(p["IMAGE"]) == 1 and p["IMAGE"][0].size == (64, 64)
        assert len(p["AUDIO"]) == 1 and p["AUDIO"][0].shape == (80000,) and np.isfinite(p["AUDIO"][0]).all()
        assert len(p["VIDEO"]) == 1 and len(p["VIDEO"][0]) == 16 and p["VIDEO"][0][0].shape == (320, 576, 3)
    # different captions -> different samples (the batch rows are not copies of one another)
    assert not np.array_equal(np.asarray(outs[0][1]["IMAGE"][0]), np.asarray(outs[1][1]["IMAGE"][0]))
    # second call: the pipelines replay the graphs the first call captured
    outs2 = dec.generate_batch([{"llm_text_all": [t]} for t in texts])
    assert calls == dict(IMAGE=2, AUDIO=2, VIDEO=2) and len(outs2) == 3
    for i, (a, p, pt) in enumerate(outs2):
        assert pt["VIDEO"] == [f"sea {i}"] and len(p["VIDEO"]) == 1 and len(p["VIDEO"][0]) == 16 and p["VIDEO"][0][0].shape == (320, 576, 3)
        assert len(p["IMAGE"]) == 1 and p["IMAGE"][0].size == (64, 64) and p["AUDIO"][0].shape == (80000,) and np.isfinite(p["AUDIO"][0]).all()
        assert all(np.isfinite(np.asarray(f, dtype=np.float32)).all() for f in p["VIDEO"][0])


def _close_np(a, b, tol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    e = float(np.abs(a - b).mean() / (np.abs(b).mean() + 1e-12))
    assert e < tol, f"{what}: mean relative difference {e:.3e}"


def test_sd_pipeline_rows_of_a_batch_equal_single_calls(dev):
    """Row b of a batched call = the call on prompt b alone with latent b (other tilings of the same arithmetic: a bound, not bit equality),
    `num_images_per_prompt` = the prompt repeated, guidance 1.0 runs without the unconditional half, a per-prompt negative prompt list is
    honoured, and the reference's argument checks raise (custom_sd.py:395-457)."""
    pipe, _ = _pipe(dev)
    prompts = ["a red car", "a blue boat on a lake", "x"]
    lat = torch.randn(3, 4, 16, 24, generator=torch.Generator().manual_seed(3))
    kw = dict(num_inference_steps=5, guidance_scale=6.0, height=64, width=96, output_type="np")
    batched = pipe(prompt=prompts, latents=lat, **kw).images
    assert batched.shape == (3, 64, 96, 3)
    for b, p in enumerate(prompts):
        single = pipe(prompt=[p], latents=lat[b:b + 1], **kw).images
        _close_np(batched[b], single[0], 2e-2, f"row {b}")
    two = pipe(prompt=["a red car"], num_images_per_prompt=2, latents=lat[:2], **kw).images
    rep = pipe(prompt=["a red car", "a red car"], latents=lat[:2], **kw).images
    assert np.array_equal(two, rep)
    assert not np.array_equal(two[0], two[1])                      # two latents -> two images
    neg = pipe(prompt=prompts[:2], negative_prompt=["blurry", "dark"], latents=lat[:2], **kw).images
    neg_single = pipe(prompt=prompts[1:2], negative_prompt=["dark"], latents=lat[1:2], **kw).images
    _close_np(neg[1], neg_single[0], 2e-2, "negative prompt row")
    assert np.abs(neg[1] - batched[1]).mean() > 1e-4               # the negative prompt does change the sample
    nocfg = pipe(prompt=prompts[:1], latents=lat[:1], num_inference_steps=5, guidance_scale=1.0, height=64, width=96, output_type="np").images
    assert nocfg.shape == (1, 64, 96, 3) and np.isfinite(nocfg).all() and np.abs(nocfg[0] - batched[0]).mean() > 1e-4
    with pytest.raises(ValueError):
        pipe(prompt=prompts[:1], height=60, width=96)                              # not divisible by 8
    with pytest.raises(ValueError):
        pipe(prompt=prompts[:2], negative_prompt=["only one"], **kw)               # negative prompt batch mismatch
    with pytest.raises(ValueError):
        pipe(prompt=prompts[:1], prompt_embeds=torch.zeros(1, 77, 64), **kw)       # both prompt and prompt_embeds
    with pytest.raises(ValueError):
        pipe(**kw)                                                                 # neither


def test_audio_and_video_pipeline_rows_of_a_batch_equal_single_calls(dev):
    from helpers import tiny_audio_pipe, tiny_video_pipe
    ad = tiny_audio_pipe(dev)
    prompts = ["rain on a tin roof", "a dog barks twice"]
    g = lambda s: torch.Generator(device=dev).manual_seed(s)
    both = ad(prompt=prompts, num_inference_steps=4, audio_length_in_s=0.5, generator=g(1)).audios
    assert both.shape[0] == 2 and np.isfinite(both).all() and not np.array_equal(both[0], both[1])
    two = ad(prompt=prompts[:1], num_waveforms_per_prompt=2, num_inference_steps=4, audio_length_in_s=0.5, generator=g(1)).audios
    assert two.shape == both.shape
    with pytest.raises(ValueError):
        ad(prompt=prompts, audio_length_in_s=-1.0)
    vd = tiny_video_pipe(dev)
    lat = torch.randn(2, 4, 3, 8, 8, generator=torch.Generator().manual_seed(7))
    kw = dict(num_frames=3, num_inference_steps=4, height=32, width=32, output_type="pt")
    vb = vd(prompt=["waves at sunset", "a city at night"], latents=lat, **kw).frames
    assert vb.shape[0] == 2 and torch.isfinite(vb).all()
    for b, p in enumerate(["waves at sunset", "a city at night"]):
        vs = vd(prompt=[p], latents=lat[b:b + 1], **kw).frames
        _close_np(vb[b].float().cpu().numpy(), vs[0].float().cpu().numpy(), 3e-2, f"video row {b}")
    with pytest.raises(ValueError):
        vd(prompt=["x"], height=30, width=32)
