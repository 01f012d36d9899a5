"""Debug aid: per-block relative error of the UNet engine against the oracle (inputs taken from the oracle)."""
import torch
from oracle.unet import UNetCfg, UNetOracle, random_unet_weights
from spider_amd.unet import UNetConfig, UNetEngine
dev = torch.device("cuda:0")
BF = torch.bfloat16
for sdxl in (False, True):
    ocfg = UNetCfg.tiny(sdxl)
    w = random_unet_weights(ocfg, seed=1)
    orc = UNetOracle(ocfg, w)
    eng = UNetEngine(UNetConfig(**ocfg.__dict__), w, dev)
    g = torch.Generator().manual_seed(2)
    B2 = 2
    x = torch.randn(B2, 4, 16, 24, generator=g).bfloat16().float()
    enc = torch.randn(B2, 77, ocfg.cross_dim, generator=g).bfloat16().float()
    added = None
    if sdxl:
        added = dict(text_embeds=torch.randn(B2, 64, generator=g).bfloat16().float(),
                     time_ids=torch.tensor([[128, 192, 0, 0, 128, 192]] * B2, dtype=torch.float32))
    t = torch.tensor(500)
    eng.prepare(torch.tensor([500]), enc.to(dev), added)
    eng.tproj_cur.copy_(eng.tproj_steps[0])
    temb = orc.time_embed(t, B2, added)
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    nhwc = lambda z: z.permute(0, 2, 3, 1).contiguous().to(dev).to(BF)
    nchw = lambda z: z.permute(0, 3, 1, 2)
    # time projection of first resnet
    r0 = "down_blocks.0.resnets.0"
    tp_ref = orc._lin(r0 + ".time_emb_proj", torch.nn.functional.silu(temb))
    print("sdxl", sdxl, "tproj", rel(eng.tproj_view[r0], tp_ref))
    h = orc._conv("conv_in", x)
    from spider_amd import ops
    print(" conv_in", rel(nchw(ops.conv2d_small_cin(nhwc(x), eng.w["conv_in.weight"], eng.w["conv_in.bias"])), h))
    hb = h.bfloat16().float()
    ref = orc.resnet(r0, hb, temb)
    print(" resnet", rel(nchw(eng._resnet(r0, nhwc(hb))), ref))
    # pieces of the resnet
    a_ref = torch.nn.functional.silu(orc._gn(r0 + ".norm1", hb))
    print("  gn+silu", rel(nchw(eng._gn(r0 + ".norm1", nhwc(hb), True)), a_ref))
    ab = a_ref.bfloat16().float()
    c_ref = orc._conv(r0 + ".conv1", ab)
    print("  conv1", rel(nchw(ops.conv2d(nhwc(ab), eng.w[r0 + ".conv1.weight"], bias=eng.w[r0 + ".conv1.bias"])), c_ref))
    ti = 1 if not ocfg.down_attn[0] else 0
    tn = f"down_blocks.{ti}.attentions.0"
    ci = ocfg.block_out[ti]
    z = torch.randn(B2, ci, 8, 12, generator=g).bfloat16().float()
    ref = orc.transformer(tn, z, enc, ocfg.heads[ti], ocfg.depth[ti])
    print(" transformer", rel(nchw(eng._transformer(tn, nhwc(z), ocfg.heads[ti], ocfg.depth[ti])), ref))
    # attention alone
    b = tn + ".transformer_blocks.0"
    y = torch.randn(B2, 96, ci, generator=g).bfloat16().float()
    ref = orc.attention(b + ".attn1", y, y, ocfg.heads[ti])
    o = eng._self_attn(b, y.to(dev).to(BF), ocfg.heads[ti])
    got = ops.gemm(o, eng.w[b + ".attn1.to_out.0.weight"], bias=eng.w[b + ".attn1.to_out.0.bias"])
    print("  self-attn", rel(got, ref))
    full = orc.forward(x, t, enc, added)
    got = eng.step(nhwc(x), 0, use_graph=False)
    print(" full", rel(nchw(got), full))
