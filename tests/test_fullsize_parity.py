"""GPU: engine-vs-oracle comparisons AT BASELINE SIZES (VERDICT r2 "weak" #4): the dispatch heuristics, folded LayerNorm, fused
cross-attention, GroupNorm-cat, split-K choices and graph capture at the real 320 / 640 / 1280 widths and 3584 / 4096-wide LLM
layers are checked against the fp32 CPU oracle, not only against themselves (tests/test_fullsize_properties.py).

  * SD-v1.5 UNet (859.5 M parameters) at [2,4,64,64] / [2,77,768]: the configs[1] decoder step   (custom_sd.py:634-639)
  * SDXL UNet (2.57 B parameters) at 64^2, CFG batch 2                                             (Comic_Generation.py:440)
  * AudioLDM-L UNet (739 M) at [2,8,125,16] with class-label conditioning                          (custom_ad.py:575-581)
  * zeroscope UNet3D (1.41 B) at 2 frames of 40 x 72                                                (custom_vd.py:671-676)
  * Qwen2.5-Omni-7B thinker / Llama-8B decoder layers at full width (2 layers): prefill of 300 tokens + 4 decode steps
    (modeling_llama3.py:202-361,576-631,854-887)

Every UNet comparison runs in BOTH engine dtypes: bf16 (BASELINE's stated dtype) and f16 (the reference's torch_dtype,
spider_decoder.py:109). Bounds = value measured on MI355X + 20 % (printed as MEASURED ... under -s); DESIGN.md section 4 holds
the table next to north_star's 1e-3. The oracle forward passes take 1-10 s each on the box's host cores."""
import gc
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f16": torch.float16}
# one UNet evaluation, rel-L2 against the fp32 oracle; measured on MI355X (round 3) + 20 %
BOUND = {
    "sd15": {"bf16": 1.6e-2, "f16": 2.0e-3},             # measured 1.305e-2 / 1.64e-3
    "sd15_s32": {"bf16": 1.2e-2, "f16": 1.45e-3},        # fp32 residual stream: measured 0.977e-2 / 1.21e-3 (a fall-back to the 16-bit stream, 1.64e-3, fails)
    "sdxl_s32": {"bf16": 1.65e-2, "f16": 1.9e-3},        # SDXL with the fp32 stream, as story.py loads it: measured 1.362e-2 / 1.56e-3
    "zeroscope_s32": {"bf16": 1.3e-2, "f16": 1.6e-3},    # fp32 stream (TextToVideoSDPipeline.from_pretrained's mode): measured 1.064e-2 / 1.34e-3
    "sdxl": {"bf16": 2.0e-2, "f16": 2.4e-3},             # 1.637e-2 / 1.97e-3
    "audioldm_l": {"bf16": 1.4e-2, "f16": 1.75e-3},      # 1.154e-2 / 1.43e-3
    "zeroscope": {"bf16": 1.9e-2, "f16": 2.3e-3},        # 1.541e-2 / 1.92e-3
}


def _rel(a, b):
    return float((a.float().cpu() - b).norm() / b.norm())


def _free():
    gc.collect()
    torch.cuda.empty_cache()


_WEIGHTS = {}


def _weights(kind: str, seed: int):
    """Random fp32 weights of a full-size model, made once per (model, seed): generating SDXL's 2.57 B / zeroscope's 1.41 B parameters
    costs 15-25 s of host time, and five tests of this module use the same set (every fixture was generated with that seed). One set is
    kept at a time (the tests are ordered by model); engines copy what they need to the GPU and leave the host tensors alone."""
    key = (kind, int(seed))
    if key not in _WEIGHTS:
        _WEIGHTS.clear()
        gc.collect()
        if kind == "zeroscope":
            from oracle.unet3d import UNet3DCfg, random_unet3d_weights
            _WEIGHTS[key] = random_unet3d_weights(UNet3DCfg.zeroscope(), seed=seed)
        else:
            from oracle.unet import UNetCfg, random_unet_weights
            _WEIGHTS[key] = random_unet_weights(getattr(UNetCfg, kind)(), seed=seed)
    return _WEIGHTS[key]


@pytest.fixture(scope="module")
def sd15_case():
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    ocfg = UNetCfg.sd15()
    w = random_unet_weights(ocfg, seed=0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 64, 64, generator=g).bfloat16().float()
    enc = torch.randn(2, 77, 768, generator=g).bfloat16().float()
    t = torch.tensor(500)
    ref = UNetOracle(ocfg, w).forward(x, t, enc)
    yield ocfg, w, x, enc, t, ref
    del w
    _free()


@pytest.mark.parametrize("stream32", [False, True])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_sd15_unet_step_fullsize_matches_oracle(dev, sd15_case, dtype, stream32):
    """The 2 x 2 precision table of DESIGN.md section 4 at the configs[1] size: {bf16, f16} x {16-bit, fp32 residual stream}."""
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg, w, x, enc, t, ref = sd15_case
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype], stream32=stream32)
    eng.prepare(torch.tensor([int(t)]), enc.to(dev))
    assert len(eng.xf) > 0, "the SD-v1.5 64^2 / 32^2 sites must take the fused cross-attention path"
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev).to(DT[dtype])
    eager = eng.step(xn, 0, use_graph=False).permute(0, 3, 1, 2).clone()
    graph = eng.step(xn, 0, use_graph=True).permute(0, 3, 1, 2)
    assert torch.equal(eager, graph), "hipGraph replay must be bit-identical to eager launches"
    r = _rel(eager, ref)
    print(f"MEASURED fullsize sd15 unet_step dtype={dtype} stream32={stream32} rel={r:.5f}")
    assert r < BOUND["sd15_s32" if stream32 else "sd15"][dtype], r
    del eng
    _free()


PRECISE_BOUND = 1.0e-3          # north_star's bound itself
# measured on MI355X (round 5), one evaluation / whole loop: SD-v1.5 5.4e-4 / 5.7e-4 (level 1), 4.0e-4 (level 2); SDXL 1.27e-3 (level 1:
# its error sits in the 70 transformer blocks), 7.7e-4 / 6.5e-4 (level 2); zeroscope 8.0e-4 (level 1), 5.7e-4 / 8.0e-4 (level 2)
SDXL_PRECISE_BOUND = {1: 1.55e-3, 2: PRECISE_BOUND}
ZS_PRECISE_BOUND = {1: 1.25e-3, 2: PRECISE_BOUND}


@pytest.mark.parametrize("level", [1, 2])
def test_sd15_unet_step_fullsize_precise_mode_inside_1e3(dev, sd15_case, level):
    """UNetEngine(precise=True), f16: every read of the residual stream on its fp32 master (hi / lo operand split, fp32 GroupNorm
    inputs), GEGLU rounded once -- one evaluation at the configs[1] size inside north_star's 1e-3 of the fp32 oracle."""
    import time
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg, w, x, enc, t, ref = sd15_case
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=torch.float16, precise=level)
    eng.prepare(torch.tensor([int(t)]), enc.to(dev))
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)                 # fp32 NHWC: the precise conv_in reads the un-rounded latents
    eager = eng.step(xn, 0, use_graph=False).permute(0, 3, 1, 2).clone()
    graph = eng.step(xn, 0, use_graph=True).permute(0, 3, 1, 2)
    assert torch.equal(eager, graph), "hipGraph replay must be bit-identical to eager launches"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.step(xn, 0, use_graph=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    r = _rel(eager, ref)
    print(f"MEASURED fullsize sd15 unet_step precise={level} f16 rel={r:.5f} ms_per_step={ms:.3f}")
    assert r < PRECISE_BOUND, r
    del eng
    _free()


# latents after the FULL denoising loop of configs[1] -- the quantity north_star names ("bf16 UNet latents agree within 1e-3
# relative"): 40 PNDM steps = 41 UNet evaluations at CFG batch 2, guidance 7.5, [1,4,64,64] (custom_sd.py:627-652), engine in the
# pipelines' default mode (f16 + fp32 residual stream) against the fp32 oracle loop (41 x 2.3 s of oracle on the box's host cores).
LOOP41_BOUND = 1.55e-3         # measured on MI355X (round 4): 1.26e-3 -- the loop does not amplify the per-evaluation 1.21e-3


def _loop_fixture(golden_dir, name):
    """an fp32 oracle loop run once in the build container (tests/golden/make_oracle_loops.py): the GPU suite spends its time on the
    engines, not on minutes of host arithmetic"""
    import numpy as np
    return np.load(os.path.join(golden_dir, name))


def test_sd15_full_41_step_loop_latents_match_oracle(dev, sd15_case, golden_dir):
    import numpy as np
    from spider_amd.schedulers import PNDMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine, denoise
    ocfg, w, x, enc, t, ref1 = sd15_case
    fx = _loop_fixture(golden_dir, "oracle_loop_sd15_pndm40.npz")
    lat = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(11))
    assert np.array_equal(lat.numpy(), fx["latents_in"]) and abs(float(enc.double().sum()) - float(fx["enc_sum"])) < 1e-6
    assert int(fx["steps"]) == 40 and float(fx["guidance"]) == 7.5 and int(fx["weights_seed"]) == 0
    ref = torch.from_numpy(fx["latents_out"])          # = denoise_loop(UNetOracle(ocfg, w), PNDMOracle(), lat, enc, 7.5, 40)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=torch.float16, stream32=True)
    got = denoise(eng, PNDMScheduler(), lat.to(dev), enc.to(dev), 7.5, 40)
    r = _rel(got, ref)
    # how far the loop moved the latents at all (a loop that returned its input would score this)
    moved = float((ref - lat).norm() / ref.norm())
    print(f"MEASURED fullsize sd15 41-step PNDM loop latents f16+stream32 rel={r:.5f} (loop displacement {moved:.3f})")
    assert moved > 0.05
    assert r < LOOP41_BOUND, r
    del eng
    _free()


def test_sd15_full_41_step_loop_latents_precise_mode_inside_1e3(dev, sd15_case, golden_dir):
    """the same loop with UNetEngine(precise=True): north_star's quantity inside north_star's bound"""
    import numpy as np
    from spider_amd.schedulers import PNDMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine, denoise
    ocfg, w, x, enc, t, ref1 = sd15_case
    fx = _loop_fixture(golden_dir, "oracle_loop_sd15_pndm40.npz")
    lat = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(11))
    assert np.array_equal(lat.numpy(), fx["latents_in"])
    ref = torch.from_numpy(fx["latents_out"])
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=torch.float16, precise=True)
    got = denoise(eng, PNDMScheduler(), lat.to(dev), enc.to(dev), 7.5, 40)
    r = _rel(got, ref)
    print(f"MEASURED fullsize sd15 41-step PNDM loop latents precise f16 rel={r:.5f}")
    assert r < PRECISE_BOUND, r
    del eng
    _free()


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_audioldm_l_unet_step_fullsize_matches_oracle(dev, dtype):
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg = UNetCfg.audioldm_l()
    w = random_unet_weights(ocfg, seed=2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 125, 16, generator=g).bfloat16().float()
    cl = torch.nn.functional.normalize(torch.randn(2, ocfg.class_in, generator=g), dim=-1).bfloat16().float()
    t = torch.tensor(601)
    ref = UNetOracle(ocfg, w).forward(x, t, None, None, cl)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype])
    eng.prepare(torch.tensor([601]), None, class_labels=cl.to(dev))
    got = eng.step(x.permute(0, 2, 3, 1).contiguous().to(dev).to(DT[dtype]), 0, use_graph=True).permute(0, 3, 1, 2)
    r = _rel(got, ref)
    print(f"MEASURED fullsize audioldm_l unet_step dtype={dtype} rel={r:.5f}")
    assert r < BOUND["audioldm_l"][dtype], r
    del eng, w
    _free()


def test_audioldm_l_full_40_step_loop_latents_match_oracle(dev, golden_dir):
    """configs[3]/[4]'s audio decoder at full size: the whole AudioLDM denoising loop (custom_ad.py:568-594: CFG batch 2, class-label
    conditioning, no encoder_hidden_states; 40 DDIM steps, guidance 2.5) on the `[1, 8, 125, 16]` latent of 5 s of audio, engine in the
    mode AudioLDMPipeline.from_pretrained loads (f16 + fp32 residual stream), against the fp32 oracle loop."""
    import numpy as np
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine, denoise
    ocfg = UNetCfg.audioldm_l()
    w = random_unet_weights(ocfg, seed=2)
    g = torch.Generator().manual_seed(13)
    lat = torch.randn(1, 8, 125, 16, generator=g)
    cl = torch.nn.functional.normalize(torch.randn(2, ocfg.class_in, generator=g), dim=-1).bfloat16().float()
    fx = _loop_fixture(golden_dir, "oracle_loop_audioldm_l_ddim40.npz")
    assert np.array_equal(lat.numpy(), fx["latents_in"]) and np.array_equal(cl.numpy(), fx["class_labels"])
    assert int(fx["steps"]) == 40 and float(fx["guidance"]) == 2.5 and int(fx["weights_seed"]) == 2
    ref = torch.from_numpy(fx["latents_out"])   # = denoise_loop(UNetOracle(ocfg, w), DDIMOracle(), lat, None, 2.5, 40, class_labels=cl)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=torch.float16, stream32=True)
    got = denoise(eng, DDIMScheduler(), lat.to(dev), None, 2.5, 40, class_labels=cl.to(dev))
    r = _rel(got, ref)
    moved = float((ref - lat).norm() / ref.norm())
    print(f"MEASURED fullsize audioldm_l 40-step DDIM loop latents f16+stream32 rel={r:.5f} (loop displacement {moved:.3f})")
    assert moved > 0.05
    assert r < 4.5e-4, r          # measured 3.6e-4 on MI355X (round 4): the audio latents are INSIDE north_star's 1e-3
    del eng, w
    _free()


@pytest.fixture(scope="module")
def sdxl_case():
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    ocfg = UNetCfg.sdxl()
    w = _weights("sdxl", 4)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 64, 64, generator=g).bfloat16().float()
    enc = torch.randn(2, 77, 2048, generator=g).bfloat16().float()
    added = dict(text_embeds=torch.randn(2, 1280, generator=g).bfloat16().float(),
                 time_ids=torch.tensor([[512, 512, 0, 0, 512, 512]] * 2, dtype=torch.float32))
    t = torch.tensor(441)
    ref = UNetOracle(ocfg, w).forward(x, t, enc, added)
    yield ocfg, w, x, enc, added, t, ref
    del w
    _free()


@pytest.mark.parametrize("stream32", [False, True])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_sdxl_unet_step_fullsize_matches_oracle(dev, sdxl_case, dtype, stream32):
    """stream32=True is the mode story.py loads the SDXL engine in (init_story_generation)."""
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg, w, x, enc, added, t, ref = sdxl_case
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype], stream32=stream32)
    eng.prepare(torch.tensor([int(t)]), enc.to(dev), added)
    got = eng.step(x.permute(0, 2, 3, 1).contiguous().to(dev).to(DT[dtype]), 0, use_graph=True).permute(0, 3, 1, 2)
    r = _rel(got, ref)
    print(f"MEASURED fullsize sdxl unet_step dtype={dtype} stream32={stream32} rel={r:.5f}")
    assert r < BOUND["sdxl_s32" if stream32 else "sdxl"][dtype], r
    del eng
    _free()


@pytest.mark.parametrize("level", [1, 2])
def test_sdxl_unet_step_fullsize_precise_mode(dev, sdxl_case, level):
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg, w, x, enc, added, t, ref = sdxl_case
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=torch.float16, precise=level)
    eng.prepare(torch.tensor([int(t)]), enc.to(dev), added)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    eager = eng.step(xn, 0, use_graph=False).permute(0, 3, 1, 2).clone()
    got = eng.step(xn, 0, use_graph=True).permute(0, 3, 1, 2)
    assert torch.equal(eager, got)
    r = _rel(got, ref)
    print(f"MEASURED fullsize sdxl unet_step precise={level} f16 rel={r:.5f}")
    assert r < SDXL_PRECISE_BOUND[level], r
    del eng
    _free()


# measured on MI355X (round 4): 7.1e-4 / 1.52e-3 / 1.69e-3 / 1.69e-3 -- the loop settles at the per-evaluation error (1.50 - 1.56e-3)
SDXL_LOOP50_BOUND = {"after_1": 8.6e-4, "after_10": 1.85e-3, "after_25": 2.05e-3, "latents_out": 2.05e-3}


@pytest.mark.parametrize("precise", [0, 2])
def test_sdxl_full_50_step_ddim_loop_latents_match_oracle_fixture(dev, golden_dir, precise):
    """The story decoder's loop settings at full size (SDXL UNet, 50 DDIM steps, guidance 5.0, CFG batch 2 on a [1, 4, 64, 64] latent;
    Comic_Generation.py:316-317, 440) without the consistent-self-attention coins, engine in the mode init_story_generation loads
    (f16 + fp32 residual stream). The fp32 oracle loop (50 x 6 s of host time) was run once in the build container
    (tests/golden/make_oracle_loops.py -> oracle_loop_sdxl50.npz); here only the engine runs. Checked after 1, 10, 25 and 50 steps."""
    import importlib.util
    import numpy as np
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd import ops
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine
    fx = np.load(os.path.join(golden_dir, "oracle_loop_sdxl50.npz"))
    spec = importlib.util.spec_from_file_location("make_oracle_loops", os.path.join(golden_dir, "make_oracle_loops.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    lat, enc, added = mk.sdxl_inputs()                     # the generator's own seeded inputs ...
    assert np.array_equal(lat.numpy(), fx["latents_in"]) and np.array_equal(enc.numpy(), fx["enc"])     # ... are the fixture's
    assert np.array_equal(added["text_embeds"].numpy(), fx["text_embeds"])
    ocfg = UNetCfg.sdxl()
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), _weights("sdxl", int(fx["weights_seed"])), dev, dtype=torch.float16,
                     stream32=True, precise=precise)
    sched = DDIMScheduler()
    steps, guidance = int(fx["steps"]), float(fx["guidance"])
    ts = sched.set_timesteps(steps)
    eng.prepare(ts, enc.to(dev), {k: v.to(dev) for k, v in added.items()})
    x = (lat.to(dev) * sched.init_noise_sigma).contiguous()
    rels = {}
    for i, t in enumerate(ts):                             # spider_amd.unet.denoise, unrolled to look at intermediate latents
        xin = ops.latent_to_nhwc_f32(x, reps=2) if precise else ops.latent_to_nhwc(x, reps=2, dtype=eng.dtype)
        e = eng.step(xin, i, use_graph=True)
        x = sched.step(ops.cfg_combine(e, guidance), t, x)
        name = "latents_out" if i + 1 == steps else f"after_{i + 1}"
        if name in fx.files:
            rels[name] = _rel(x, torch.from_numpy(fx[name]))
    moved = float(np.linalg.norm(fx["latents_out"] - fx["latents_in"]) / np.linalg.norm(fx["latents_out"]))
    print(f"MEASURED fullsize sdxl 50-step DDIM loop latents f16 {'precise=%d' % precise if precise else 'stream32'} " +
          " ".join(f"{k}={v:.5f}" for k, v in rels.items()) + f" (loop displacement {moved:.3f})")
    assert moved > 0.05 and set(rels) == set(SDXL_LOOP50_BOUND)
    for k, v in rels.items():
        assert v < (SDXL_PRECISE_BOUND[precise] if precise else SDXL_LOOP50_BOUND[k]), (k, v, rels)
    del eng
    _free()


STORY_BOUND = {0: 2.2e-3, 2: 1.0e-3}      # f16 + fp32 stream (the mode init_story_generation loads): measured 1.52 / 1.66e-3; precise=2: north_star's bound


@pytest.mark.parametrize("precise", [0, 2])
def test_story_sdxl_fullsize_write_then_read_matches_oracle_fixture(dev, golden_dir, precise):
    """BASELINE configs[2]'s actual arithmetic at a BASELINE size (VERDICT r4 missing #2): SDXL UNet at the 64^2 latent, CFG batch 8,
    FreeU(0.6, 0.4, 1.1, 1.2), consistent self-attention on the 36 up-block processors in its masked branch at cur_step = 5 -- the WRITE
    evaluation (Comic_Generation.py:94-118, 440), then the READ evaluation at CFG batch 2 against the bank it left (:104-109) --
    HIP engine in the mode init_story_generation loads (f16 + fp32 residual stream) against the fp32 oracle fixture generated in the
    build container (tests/golden/make_story_fullsize.py; inputs regenerated from its seeds and checked against its checksums)."""
    import importlib.util
    import numpy as np
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.story import ConsistentSelfAttention, StoryState
    from spider_amd.unet import UNetConfig, UNetEngine
    fx = np.load(os.path.join(golden_dir, "oracle_story_sdxl_fullsize.npz"))
    spec = importlib.util.spec_from_file_location("make_story_fullsize", os.path.join(golden_dir, "make_story_fullsize.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    inp = mk.story_inputs()
    assert abs(float(inp["write"]["x"].double().sum()) - float(fx["x_write_sum"])) < 1e-6
    assert abs(float(inp["read"]["x"].double().sum()) - float(fx["x_read_sum"])) < 1e-6
    ocfg = UNetCfg.sdxl()
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), _weights("sdxl", int(fx["weights_seed"])), dev, dtype=torch.float16, stream32=True,
                     precise=precise)
    eng.freeu = tuple(float(v) for v in fx["freeu"])
    ui = iter(inp["uniforms"])
    st = StoryState(total_count=ConsistentSelfAttention.count_processors(eng), height=mk.HH, width=mk.WW, coin=lambda: 0.95,
                    uniforms=lambda n: next(ui)[:n])
    assert st.total_count == 36
    st.regen_masks(dev)
    eng.self_attn_hook = ConsistentSelfAttention(st)
    rels = {}
    for phase in ("write", "read"):
        c = inp[phase]
        st.write, st.cur_step, st.attn_count = phase == "write", int(fx["step"]), 0
        eng.prepare(torch.tensor([int(c["t"])]), c["enc"].to(dev), dict(text_embeds=c["text_embeds"], time_ids=c["time_ids"]))
        xin = c["x"].permute(0, 2, 3, 1).contiguous().to(dev)
        e = eng.step(xin if precise else xin.half(), 0)          # precise: the un-rounded fp32 latents, as unet.denoise hands them over
        rels[phase] = _rel(e.permute(0, 3, 1, 2), torch.from_numpy(fx[phase]))
    eff = float(fx["consistent_effect"])
    print(f"MEASURED fullsize story sdxl (CFG 8, FreeU, consistent SA) f16+stream32 precise={precise} write={rels['write']:.5f} read={rels['read']:.5f} "
          f"(the masked attention moves the write evaluation by {eff:.3f})")
    assert eff > 10 * max(rels.values()), "the fixture must tell the consistent path from plain attention"
    for k, v in rels.items():
        assert v < STORY_BOUND[precise], (k, v)          # the plain SDXL evaluation measures 1.50 - 1.56e-3 at level 0 (+ the masked attention sites)
    del eng
    _free()


@pytest.fixture(scope="module")
def zeroscope_case():
    from oracle.unet3d import UNet3DCfg, UNet3DOracle, random_unet3d_weights
    ocfg = UNet3DCfg.zeroscope()
    w = _weights("zeroscope", 6)
    g = torch.Generator().manual_seed(7)
    frames = 2
    x = torch.randn(2, 4, frames, 40, 72, generator=g).bfloat16().float()
    enc = torch.randn(2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    ref = UNet3DOracle(ocfg, w).forward(x, torch.tensor(701), enc)          # one oracle evaluation for the four engine modes
    yield ocfg, w, x, enc, frames, ref
    del w
    _free()


@pytest.mark.parametrize("stream32", [False, True])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_zeroscope_unet3d_step_fullsize_matches_oracle(dev, zeroscope_case, dtype, stream32):
    from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
    ocfg, w, x, enc, frames, ref = zeroscope_case
    eng = UNet3DEngine(UNet3DConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype], stream32=stream32)
    eng.prepare(torch.tensor([701]), enc.to(dev), frames=frames)
    B, C, F_, H, W = x.shape
    xn = x.permute(0, 2, 3, 4, 1).reshape(B * F_, H, W, C).contiguous().to(dev).to(DT[dtype])
    y = eng.step(xn, 0, use_graph=True)
    got = y.view(B, F_, H, W, -1).permute(0, 4, 1, 2, 3)
    r = _rel(got, ref)
    print(f"MEASURED fullsize zeroscope unet3d_step dtype={dtype} stream32={stream32} rel={r:.5f}")
    assert r < BOUND["zeroscope_s32" if stream32 else "zeroscope"][dtype], r
    del eng
    _free()


@pytest.mark.parametrize("level", [1, 2])
def test_zeroscope_unet3d_step_precise_mode_inside_1e3(dev, zeroscope_case, level):
    from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
    ocfg, w, x, enc, frames, ref = zeroscope_case
    eng = UNet3DEngine(UNet3DConfig(**ocfg.__dict__), w, dev, dtype=torch.float16, precise=level)
    eng.prepare(torch.tensor([701]), enc.to(dev), frames=frames)
    B, C, F_, H, W = x.shape
    xn = x.permute(0, 2, 3, 4, 1).reshape(B * F_, H, W, C).contiguous().to(dev)
    eager = eng.step(xn, 0, use_graph=False).clone()
    y = eng.step(xn, 0, use_graph=True)
    assert torch.equal(eager, y)
    r = _rel(y.view(B, F_, H, W, -1).permute(0, 4, 1, 2, 3), ref)
    print(f"MEASURED fullsize zeroscope unet3d_step precise={level} f16 rel={r:.5f}")
    assert r < PRECISE_BOUND, r
    del eng
    _free()


# measured: 8 frames 1.34e-3, 16 frames 1.24e-3. The oracle's 8-frame evaluation is a committed fixture since round 6 (tests/golden/
# make_oracle_loops.py zeroscope8: the live oracle cost 50-65 s of the suite's time budget per run); the full 16 frames are compared over
# the whole loop against the committed oracle-loop fixture (test_zeroscope_full_40_step_loop_latents_match_oracle_fixture).
@pytest.mark.parametrize("frames", [8])
def test_zeroscope_unet3d_step_full_frames_matches_oracle(dev, golden_dir, frames):
    """The video decoder at 8 of the 16 frames of 40 x 72 that configs[3]/[4] decode (custom_vd.py:671-676 with num_frames=16,
    spider_decoder.py:122), in the mode TextToVideoSDPipeline.from_pretrained loads (f16 + fp32 residual stream): the temporal convs and
    the frame attention see a real frame axis (the 2-frame cases above leave the (3,1,1) convs two thirds zero padding). Oracle output
    from the fixture; the inputs are regenerated from its seeds and checked against its checksums."""
    import importlib.util
    import numpy as np
    from oracle.unet3d import UNet3DCfg
    from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
    fx = np.load(os.path.join(golden_dir, "oracle_step_zeroscope_8frames.npz"))
    spec = importlib.util.spec_from_file_location("make_oracle_loops", os.path.join(golden_dir, "make_oracle_loops.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert int(fx["frames"]) == frames
    x, enc = mk.zeroscope8_inputs(frames)
    assert abs(float(x.double().sum()) - float(fx["x_sum"])) < 1e-6 and abs(float(enc.double().sum()) - float(fx["enc_sum"])) < 1e-6
    ref = torch.from_numpy(fx["ref"])
    ocfg = UNet3DCfg.zeroscope()
    eng = UNet3DEngine(UNet3DConfig(**ocfg.__dict__), _weights("zeroscope", int(fx["weights_seed"])), dev, dtype=torch.float16, stream32=True)
    eng.prepare(torch.tensor([int(fx["t"])]), enc.to(dev), frames=frames)
    B, C, F_, H, W = x.shape
    xn = x.permute(0, 2, 3, 4, 1).reshape(B * F_, H, W, C).contiguous().to(dev).to(torch.float16)
    got = eng.step(xn, 0, use_graph=True).view(B, F_, H, W, -1).permute(0, 4, 1, 2, 3)
    r = _rel(got, ref)
    print(f"MEASURED fullsize zeroscope unet3d_step {frames} frames f16+stream32 rel={r:.5f}")
    assert r < BOUND["zeroscope_s32"]["f16"], r
    del eng
    _free()


# measured on MI355X (round 4): 1.03e-3 / 1.91e-3 / 1.91e-3 (guidance 9.0 multiplies the error of eps_cond - eps_uncond by 9: per
# evaluation 1.24e-3, see test_zeroscope_unet3d_step_full_frames_matches_oracle)
ZEROSCOPE_LOOP40_BOUND = {"after_1": 1.25e-3, "after_20": 2.3e-3, "latents_out": 2.3e-3}


@pytest.mark.parametrize("precise", [0, 2])
def test_zeroscope_full_40_step_loop_latents_match_oracle_fixture(dev, golden_dir, precise):
    """configs[3] / [4]'s video decoder over its WHOLE loop at full size: zeroscope UNet3D on the [1, 4, 16, 40, 72] latent, 40 DDIM steps,
    guidance 9.0 (spider_decoder.py:122-143 -> custom_vd.py:664-697), engine in the mode TextToVideoSDPipeline.from_pretrained loads
    (f16 + fp32 residual stream). The fp32 oracle loop (40 x 65 s of host time) was run once in the build container
    (tests/golden/make_oracle_loops.py -> oracle_loop_zeroscope40_f16.npz: latents after 1, 20 and 40 steps; the seeded inputs are
    regenerated here and checked against the fixture's checksums)."""
    import importlib.util
    import numpy as np
    from oracle.unet3d import UNet3DCfg, random_unet3d_weights
    from spider_amd import ops
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
    fx = np.load(os.path.join(golden_dir, "oracle_loop_zeroscope40_f16.npz"))
    spec = importlib.util.spec_from_file_location("make_oracle_loops", os.path.join(golden_dir, "make_oracle_loops.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    frames, steps, guidance = int(fx["frames"]), int(fx["steps"]), float(fx["guidance"])
    lat, enc = mk.zeroscope_inputs(frames)
    assert abs(float(lat.double().sum()) - float(fx["latents_in_sum"])) < 1e-6 and abs(float(enc.double().sum()) - float(fx["enc_sum"])) < 1e-6
    ocfg = UNet3DCfg.zeroscope()
    eng = UNet3DEngine(UNet3DConfig(**ocfg.__dict__), _weights("zeroscope", int(fx["weights_seed"])), dev, dtype=torch.float16,
                       stream32=True, precise=precise)
    sched = DDIMScheduler()
    ts = sched.set_timesteps(steps)
    eng.prepare(ts, enc.to(dev), frames=frames)
    B, C, F_, h, w = lat.shape
    x = (lat.to(dev).permute(0, 2, 1, 3, 4).reshape(B * F_, C, h, w) * sched.init_noise_sigma).contiguous()
    rels = {}
    for i, t in enumerate(ts):                             # spider_amd.unet3d.video_denoise, unrolled to look at intermediate latents
        xin = ops.latent_to_nhwc_f32(x, reps=2) if precise else ops.latent_to_nhwc(x, reps=2, dtype=eng.dtype)
        e = eng.step(xin, i, use_graph=True)
        x = sched.step(ops.cfg_combine(e, guidance), t, x)
        name = "latents_out" if i + 1 == steps else f"after_{i + 1}"
        if name in fx.files:
            rels[name] = _rel(x.view(B, F_, C, h, w).permute(0, 2, 1, 3, 4), torch.from_numpy(fx[name]))
    moved = float(np.linalg.norm(fx["latents_out"] - lat.numpy()) / np.linalg.norm(fx["latents_out"]))
    print(f"MEASURED fullsize zeroscope 40-step DDIM loop latents f16 {'precise=%d' % precise if precise else 'stream32'} " +
          " ".join(f"{k}={v:.5f}" for k, v in rels.items()) + f" (loop displacement {moved:.3f})")
    assert moved > 0.05 and set(rels) == set(ZEROSCOPE_LOOP40_BOUND)
    for k, v in rels.items():
        assert v < (ZS_PRECISE_BOUND[precise] if precise else ZEROSCOPE_LOOP40_BOUND[k]), (k, v, rels)
    del eng
    _free()


@pytest.mark.parametrize("model", ["qwen25_7b", "llama3_8b"])
def test_llm_fullwidth_layers_match_oracle(dev, model):
    """Two full-width decoder layers + embedding + final norm + lm_head: prefill of 300 tokens (256^2 / 256x128 MFMA GEMMs, causal
    GQA flash attention), then 4 greedy decode steps on the GEMV graph. The oracle is teacher-forced with the engine's tokens (a
    random-weight model at std 0.02 has near-tied logits, so token equality is asserted in test_llm_engine.py's reference-generated
    fixtures instead): per-step logits and every hidden state within the bf16 bounds of those tests."""
    _WEIGHTS.clear()          # (the diffusion models' host weights are not needed from here on)
    gc.collect()
    import dataclasses
    from oracle.llama import LlamaCfg, LlamaOracle
    from spider_amd.llm import LlamaEngine, LLMConfig
    ocfg = dataclasses.replace(getattr(LlamaCfg, model)(), layers=2, mrope_section=None, max_pos=1024)
    w = LlamaOracle.random_weights(ocfg, seed=8, std=0.02)
    oracle = LlamaOracle(ocfg, w)
    eng = LlamaEngine(LLMConfig(**ocfg.__dict__), w, dev, max_batch=1, max_len=512)
    S, T = 300, 4
    ids = torch.randint(3, ocfg.vocab, (1, S), generator=torch.Generator().manual_seed(9))
    out = eng.generate(input_ids=ids, max_new_tokens=T, return_dict_in_generate=True, return_logits=True, output_hidden_states=True)
    seq = out.sequences.cpu()
    assert torch.equal(seq[:, :S], ids) and seq.shape[1] == S + T
    pos = torch.arange(S + T)[None]
    logits, _, hid = oracle.forward(seq, pos, None, None, all_hidden=True)           # teacher-forced on the engine's tokens
    got_steps = out.logits.float().cpu()                                             # [1, T, V]: logits that chose token S + t
    ref_steps = logits[:, S - 1:S + T - 1]
    r = float((got_steps - ref_steps).norm() / ref_steps.norm())
    print(f"MEASURED fullwidth {model} step-logits rel={r:.5f}")
    assert r < 2.1e-2, r        # measured 1.62e-2 (Qwen) / 1.71e-2 (Llama) + 20 %
    # the engine's choice is within bf16 resolution of the oracle's best logit at every step
    for t in range(T):
        tok = int(seq[0, S + t])
        gap = float(ref_steps[0, t].max() - ref_steps[0, t, tok])
        assert gap < 0.03 * float(ref_steps[0, t].abs().max()) + 1e-3, (t, gap)
    for l in range(ocfg.layers + 1):
        got = out.hidden_states[0][l].float().cpu()                                  # prompt states [1, S, H]
        rl = float((got - hid[l][:, :S]).norm() / hid[l][:, :S].norm())
        assert rl < 2e-2, (l, rl)
        for t in range(1, T):                                                        # decode-step states [1, 1, H]
            gd = out.hidden_states[t][l].float().cpu()
            rd = float((gd - hid[l][:, S + t - 1:S + t]).norm() / hid[l][:, S + t - 1:S + t].norm())
            assert rd < 2.5e-2, (l, t, rd)
    del eng, w
    _free()
