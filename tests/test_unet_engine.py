"""GPU: native UNet step / denoising loop (spider_amd/unet.py on the HIP kernels) against the fp32 CPU oracle
(oracle/unet.py -- diffusers-0.25 restatement, parity unpinned upstream) on tiny SD-v1.5-like and SDXL-like configs.

Tolerances are relative L2 errors against the fp32 oracle, derived from bf16 storage (half-ulp 2^-9 = 2.0e-3) and set at the
value measured on MI355X + 20 % (round 2; the tests print their measured value as `MEASURED ...` under -s):
  * every block (resnet / transformer / conv / attention) in isolation: measured 1.7e-3 .. 6.0e-3, bound 8e-3
    (test_unet_blocks_match_oracle)
  * a whole evaluation chains ~25 blocks -> sqrt(25) * 3e-3 ~ 1.5e-2: measured 1.44 - 1.58e-2, bound 1.9e-2 (FreeU 1.96e-2 -> 2.4e-2)
  * latents after a coarse 6-8 step loop: measured 2.4 - 2.7e-2 (each coarse step weighs eps heavily), bound 3.2e-2.
Every comparison runs in both engine dtypes: bf16 and f16 (IEEE half: the reference's own torch_dtype, spider_decoder.py:109;
half-ulp 2^-12 = 2.4e-4, i.e. 8x finer than bf16). north_star asks for 1e-3 relative on the latents: the f16 engine measures
1.5e-3 .. 2e-3 per evaluation (the reference's own graph run in torch.float16 is at 1.6 - 1.9e-3 from fp32), bf16 1.5e-2;
DESIGN.md section 4 holds the table.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


# measured on MI355X (printed by the tests as MEASURED ...) + 20 %.
# bf16: one UNet evaluation 1.44 - 1.58e-2, FreeU 1.96e-2, loops 2.4 - 2.7e-2, single blocks 1.7 - 6.0e-3
# f16 : one UNet evaluation, FreeU, loops and blocks 8x below that (round 3 measurements in DESIGN.md section 4)
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
UNET_STEP_BOUND_DT = {"bf16": 1.9e-2, "f16": 2.4e-3}      # f16 measured 1.75 - 1.98e-3
FREEU_BOUND_DT = {"bf16": 2.4e-2, "f16": 3.0e-3}          # f16 measured 2.45e-3
LOOP_BOUND_DT = {"bf16": 3.2e-2, "f16": 4.2e-3}           # f16 measured 3.13 / 3.45e-3
BLOCK_BOUND_DT = {"bf16": 8e-3, "f16": 9e-4}              # f16 measured 2.1 - 7.4e-4
UNET_STEP_BOUND = UNET_STEP_BOUND_DT["bf16"]


def _rel(a, b):
    return float((a.float().cpu() - b).norm() / b.norm())


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("sdxl_like", [False, True])
def test_unet_step_matches_oracle(dev, sdxl_like, dtype):
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg = UNetCfg.tiny(sdxl_like)
    w = random_unet_weights(ocfg, seed=1)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype])
    g = torch.Generator().manual_seed(2)
    B2, hh, ww = 2, 16, 24
    x = torch.randn(B2, 4, hh, ww, generator=g).bfloat16().float()
    enc = torch.randn(B2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    added = None
    if sdxl_like:
        added = dict(text_embeds=torch.randn(B2, 64, generator=g).bfloat16().float(),
                     time_ids=torch.tensor([[hh * 8, ww * 8, 0, 0, hh * 8, ww * 8]] * B2, dtype=torch.float32))
    oracle = UNetOracle(ocfg, w)
    ts = torch.tensor([981, 500, 21])
    eng.prepare(ts, enc.to(dev), added)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev).to(DT[dtype])
    for i, t in enumerate(ts):
        ref = oracle.forward(x, t, enc, added)
        eager = eng.step(xn, i, use_graph=False).permute(0, 3, 1, 2)
        graph = eng.step(xn, i, use_graph=True).permute(0, 3, 1, 2)
        assert torch.equal(eager.cpu(), graph.cpu()), "hipGraph replay must be bit-identical to eager launches"
        r = _rel(eager, ref)
        print(f"MEASURED unet_step dtype={dtype} sdxl_like={sdxl_like} t={int(t)} rel={r:.5f}")
        assert r < UNET_STEP_BOUND_DT[dtype], f"t={int(t)}: rel L2 {r:.4f}"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_unet_step_with_fused_cross_attention_matches_oracle(dev, monkeypatch, dtype):
    """8-head config (SD-v1.5's head count): every cross-attention sub-block runs as the ONE-launch fused kernel with the prompt's
    K / V folded into the projections (spider_xattn_fused_bf16). Against the fp32 oracle, and against the unfused HIP path."""
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg = UNetCfg.tiny8()
    w = random_unet_weights(ocfg, seed=4)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 16, 24, generator=g).bfloat16().float()
    enc = torch.randn(2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    ref = UNetOracle(ocfg, w).forward(x, torch.tensor(500), enc)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev).to(DT[dtype])
    UNET_STEP_BOUND = UNET_STEP_BOUND_DT[dtype]
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype])
    eng.prepare(torch.tensor([500]), enc.to(dev))
    assert len(eng.xf) == len(eng.cross_layers) > 0, "the 8-head config must take the fused cross-attention path"
    fused = eng.step(xn, 0, use_graph=False).permute(0, 3, 1, 2).clone()
    graph = eng.step(xn, 0, use_graph=True).permute(0, 3, 1, 2)
    assert torch.equal(fused, graph)
    monkeypatch.setenv("SPIDER_XATTN_FUSE", "0")
    eng2 = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype])
    eng2.prepare(torch.tensor([500]), enc.to(dev))
    assert len(eng2.xf) == 0
    unfused = eng2.step(xn, 0, use_graph=False).permute(0, 3, 1, 2)
    r_f, r_u, d = _rel(fused, ref), _rel(unfused, ref), _rel(fused, unfused.float().cpu())
    print(f"rel L2 vs fp32 oracle: fused {r_f:.4f}  unfused {r_u:.4f}  fused-vs-unfused {d:.4f}")
    assert r_f < UNET_STEP_BOUND and r_f < 1.3 * r_u + 0.1 * UNET_STEP_BOUND
    # a second prompt re-folds in place: same buffers (a captured graph stays valid), new result
    ptrs = {l: f["mq_fm"].data_ptr() for l, f in eng.xf.items()}
    enc2 = torch.randn(2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    eng.prepare(torch.tensor([500]), enc2.to(dev))
    assert ptrs == {l: f["mq_fm"].data_ptr() for l, f in eng.xf.items()}
    again = eng.step(xn, 0, use_graph=True).permute(0, 3, 1, 2)
    assert _rel(again, UNetOracle(ocfg, w).forward(x, torch.tensor(500), enc2)) < UNET_STEP_BOUND


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("sdxl_like", [False, True])
def test_unet_blocks_match_oracle(dev, sdxl_like, dtype):
    """Every block class in isolation (inputs rounded to bf16, taken from the oracle): one bf16 rounding of the block
    output is 2e-3 .. 4e-3 relative; the bound for a single block is 8e-3 (per-block assertions)."""
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    from spider_amd import ops
    from spider_amd.unet import UNetConfig, UNetEngine
    BF = DT[dtype]
    ocfg = UNetCfg.tiny(sdxl_like)
    w = random_unet_weights(ocfg, seed=1)
    orc = UNetOracle(ocfg, w)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=BF)
    g = torch.Generator().manual_seed(2)
    B2 = 2
    x = torch.randn(B2, 4, 16, 24, generator=g).bfloat16().float()
    enc = torch.randn(B2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    added = None
    if sdxl_like:
        added = dict(text_embeds=torch.randn(B2, 64, generator=g).bfloat16().float(),
                     time_ids=torch.tensor([[128, 192, 0, 0, 128, 192]] * B2, dtype=torch.float32))
    t = torch.tensor(500)
    eng.prepare(torch.tensor([500]), enc.to(dev), added)
    eng.tproj_cur.copy_(eng.tproj_steps[0])
    temb = orc.time_embed(t, B2, added)
    nhwc = lambda z: z.permute(0, 2, 3, 1).contiguous().to(dev).to(BF)
    nchw = lambda z: z.permute(0, 3, 1, 2)
    errs = {}
    r0 = "down_blocks.0.resnets.0"
    errs["tproj"] = _rel(eng.tproj_view[r0], orc._lin(r0 + ".time_emb_proj", torch.nn.functional.silu(temb)))
    h = orc._conv("conv_in", x)
    errs["conv_in"] = _rel(nchw(ops.conv2d_small_cin(nhwc(x), eng.w["conv_in.weight"], eng.w["conv_in.bias"])), h)
    hb = h.bfloat16().float()
    errs["resnet"] = _rel(nchw(eng._resnet(r0, nhwc(hb))), orc.resnet(r0, hb, temb))
    a_ref = torch.nn.functional.silu(orc._gn(r0 + ".norm1", hb))
    errs["gn_silu"] = _rel(nchw(eng._gn(r0 + ".norm1", nhwc(hb), True)), a_ref)
    ab = a_ref.bfloat16().float()
    errs["conv3x3"] = _rel(nchw(ops.conv2d(nhwc(ab), eng.w[r0 + ".conv1.weight"], bias=eng.w[r0 + ".conv1.bias"])),
                           orc._conv(r0 + ".conv1", ab))
    # a resnet with a channel change (1x1 shortcut conv) and the stride-2 downsampler
    r1 = "down_blocks.1.resnets.0"
    errs["resnet_shortcut"] = _rel(nchw(eng._resnet(r1, nhwc(hb))), orc.resnet(r1, hb, temb))
    dn = "down_blocks.0.downsamplers.0.conv"
    errs["downsample"] = _rel(nchw(ops.conv2d(nhwc(hb), eng.w[dn + ".weight"], bias=eng.w[dn + ".bias"], stride=2, pad=1)),
                              orc._conv(dn, hb, stride=2))
    ti = 1 if not ocfg.down_attn[0] else 0
    tn = f"down_blocks.{ti}.attentions.0"
    ci = ocfg.block_out[ti]
    z = torch.randn(B2, ci, 8, 12, generator=g).bfloat16().float()
    errs["transformer"] = _rel(nchw(eng._transformer(tn, nhwc(z), ocfg.heads[ti], ocfg.depth[ti])),
                               orc.transformer(tn, z, enc, ocfg.heads[ti], ocfg.depth[ti]))
    b = tn + ".transformer_blocks.0"
    y = torch.randn(B2, 96, ci, generator=g).bfloat16().float()
    o = eng._self_attn(b, y.to(dev).to(BF), ocfg.heads[ti])
    errs["self_attn"] = _rel(ops.gemm(o, eng.w[b + ".attn1.to_out.0.weight"], bias=eng.w[b + ".attn1.to_out.0.bias"]),
                             orc.attention(b + ".attn1", y, y, ocfg.heads[ti]))
    o = eng._cross_attn(b, y.to(dev).to(BF), ocfg.heads[ti])
    errs["cross_attn"] = _rel(ops.gemm(o, eng.w[b + ".attn2.to_out.0.weight"], bias=eng.w[b + ".attn2.to_out.0.bias"]),
                              orc.attention(b + ".attn2", y, enc, ocfg.heads[ti]))
    print(f"MEASURED per-block rel L2 dtype={dtype}:", {k: round(v, 5) for k, v in errs.items()})
    for k, v in errs.items():
        assert v < BLOCK_BOUND_DT[dtype], (k, v)


def test_unet_step_vs_bf16_reference_emulation(dev):
    """Separates 'bf16 storage' error from implementation error: the same graph evaluated with torch's CPU bf16 kernels
    (what the reference's modules would compute in bf16) differs from the fp32 oracle as much as our HIP path does, and our
    path is closer to fp32 than that emulation (fp32 accumulation, fp32 softmax / norm statistics, fewer roundings)."""
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg = UNetCfg.tiny()
    w = random_unet_weights(ocfg, seed=1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 16, 24, generator=g).bfloat16().float()
    enc = torch.randn(2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    t = torch.tensor(500)
    ref32 = UNetOracle(ocfg, w).forward(x, t, enc)
    ref16 = UNetOracle(ocfg, w, dtype=torch.bfloat16).forward(x, t, enc)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev)
    eng.prepare(torch.tensor([500]), enc.to(dev))
    got = eng.step(x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16), 0, use_graph=False).permute(0, 3, 1, 2)
    e_hip, e_emul, d = _rel(got, ref32), _rel(ref16, ref32), _rel(got, ref16)
    print(f"rel L2: hip-vs-fp32 {e_hip:.4f}  torch-bf16-vs-fp32 {e_emul:.4f}  hip-vs-torch-bf16 {d:.4f}")
    assert e_hip < UNET_STEP_BOUND
    assert e_hip < 1.5 * e_emul + 2e-3, (e_hip, e_emul)   # not worse than a bf16 run of the reference graph itself


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("sched_name,steps", [("pndm", 8), ("ddim", 6)])
def test_denoise_loop_matches_oracle(dev, sched_name, steps, dtype):
    from oracle.unet import DDIMOracle, PNDMOracle, UNetCfg, UNetOracle, denoise_loop, random_unet_weights
    from spider_amd.schedulers import DDIMScheduler, PNDMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine, denoise
    ocfg = UNetCfg.tiny()
    w = random_unet_weights(ocfg, seed=3)
    g = torch.Generator().manual_seed(4)
    lat = torch.randn(1, 4, 16, 16, generator=g)
    enc = torch.randn(2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    ref = denoise_loop(UNetOracle(ocfg, w), PNDMOracle() if sched_name == "pndm" else DDIMOracle(), lat, enc, 7.5, steps)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype])
    sched = PNDMScheduler() if sched_name == "pndm" else DDIMScheduler()
    got = denoise(eng, sched, lat.to(dev), enc.to(dev), 7.5, steps)
    assert got.shape == lat.shape
    r = _rel(got, ref)
    print(f"MEASURED denoise_loop dtype={dtype} {sched_name} rel={r:.5f}")
    assert r < LOOP_BOUND_DT[dtype], f"latents rel L2 {r:.5f}"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_freeu_matches_oracle(dev, dtype):
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg = UNetCfg.tiny(True)
    w = random_unet_weights(ocfg, seed=5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 4, 16, 16, generator=g).bfloat16().float()
    enc = torch.randn(2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    added = dict(text_embeds=torch.randn(2, 64, generator=g).bfloat16().float(),
                 time_ids=torch.tensor([[128, 128, 0, 0, 128, 128]] * 2, dtype=torch.float32))
    oracle = UNetOracle(ocfg, w); oracle.freeu = (0.6, 0.4, 1.1, 1.2)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype]); eng.freeu = (0.6, 0.4, 1.1, 1.2)
    eng.prepare(torch.tensor([300]), enc.to(dev), added)
    got = eng.step(x.permute(0, 2, 3, 1).contiguous().to(dev).to(DT[dtype]), 0, use_graph=False).permute(0, 3, 1, 2)
    ref = oracle.forward(x, torch.tensor(300), enc, added)
    r_on = _rel(got, ref)
    print(f"MEASURED freeu dtype={dtype} rel={r_on:.5f}")
    assert r_on < FREEU_BOUND_DT[dtype]
    oracle.freeu = None
    r_off = _rel(got, oracle.forward(x, torch.tensor(300), enc, added))
    assert r_off > 3 * r_on, (r_on, r_off)  # FreeU really changes the result


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_fp32_residual_stream_is_at_least_as_close(dev, dtype):
    """stream32 (fp32 master of the residual stream, 16-bit shadow for every consumer) against the plain 16-bit stream on the same
    engine dtype: closer to the fp32 oracle (the adds no longer round), graph replay still bit-identical to eager."""
    from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    rs = {}
    for sdxl_like in (False, True):
        ocfg = UNetCfg.tiny(sdxl_like)
        w = random_unet_weights(ocfg, seed=1)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(2, 4, 16, 24, generator=g).bfloat16().float()
        enc = torch.randn(2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
        added = None
        if sdxl_like:
            added = dict(text_embeds=torch.randn(2, 64, generator=g).bfloat16().float(),
                         time_ids=torch.tensor([[128, 192, 0, 0, 128, 192]] * 2, dtype=torch.float32))
        ref = UNetOracle(ocfg, w).forward(x, torch.tensor(500), enc, added)
        xn = x.permute(0, 2, 3, 1).contiguous().to(dev).to(DT[dtype])
        for s32 in (False, True):
            eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype], stream32=s32)
            eng.prepare(torch.tensor([500]), enc.to(dev), added)
            eager = eng.step(xn, 0, use_graph=False).permute(0, 3, 1, 2).clone()
            graph = eng.step(xn, 0, use_graph=True).permute(0, 3, 1, 2)
            assert torch.equal(eager, graph)
            rs[(sdxl_like, s32)] = _rel(eager, ref)
        print(f"MEASURED stream32 dtype={dtype} sdxl_like={sdxl_like}: 16-bit stream {rs[(sdxl_like, False)]:.5f}  fp32 stream {rs[(sdxl_like, True)]:.5f}")
        assert rs[(sdxl_like, True)] < rs[(sdxl_like, False)] * 1.02
        assert rs[(sdxl_like, True)] < UNET_STEP_BOUND_DT[dtype]


def test_unet_flop_table():
    from spider_amd.unet import UNetConfig, unet_flops
    f = unet_flops(UNetConfig.sd15(), 64, 64)
    # attention cores must reproduce SURVEY.md section 8d: self 122.5 GF, cross 3.56 GF per sample
    assert abs(f["attn_self"] / 1e9 - 122.5) < 0.3 and abs(f["attn_cross"] / 1e9 - 3.56) < 0.05
    assert 0.6e12 < f["total"] < 1.0e12


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("stream32", [False, True])
def test_producer_side_groupnorm_statistics_match_own_pass(dev, dtype, stream32):
    """round 4: GroupNorm statistics written by the producing conv (epilogue / split-K reduce) and Transformer2DModel.norm applied
    inside proj_in give the same evaluation as every GroupNorm running its own two passes (only the fp32 summation order of the
    statistics differs), and hipGraph replay stays bit-identical to eager launches."""
    from oracle.unet import UNetCfg, random_unet_weights
    from spider_amd.unet import UNetConfig, UNetEngine
    ocfg = UNetCfg.tiny()
    w = random_unet_weights(ocfg, seed=7)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 32, 32, 4, generator=g).to(dev).to(DT[dtype])
    enc = torch.randn(2, 77, ocfg.cross_dim, generator=g).to(dev)
    outs = {}
    for prod in (True, False):
        eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev, dtype=DT[dtype], stream32=stream32)
        eng.gn_producer, eng.gn_fuse_in = prod, prod
        eng.prepare(torch.tensor([500]), enc)
        eager = eng.step(x, 0, use_graph=False).clone()
        graph = eng.step(x, 0, use_graph=True)
        assert torch.equal(eager, graph)
        outs[prod] = eager.float()
    r = float((outs[True] - outs[False]).norm() / outs[False].norm())
    print(f"MEASURED producer-vs-own-pass GroupNorm dtype={dtype} stream32={stream32} rel={r:.6f}")
    assert r < (2.5e-2 if dtype == "bf16" else 3.2e-3), r    # two valid 16-bit executions of the graph: each is 1.5e-2 / 1.8e-3 from fp32
