"""BASELINE configs[3]/[4] (any-to-many): text -> {text + image + audio + video} through the real Decoders-Controller path
(SpiderDecoder.generate: routing + decode_image / decode_audio / decode_video, spider_decoder.py:100-166,309-348) on
random-init weights of the true shapes: Qwen2.5-7B text decoder (1536-token prompt + 128 greedy tokens), SD-v1.5 (512^2,
41 UNet calls), AudioLDM-s (5 s, 40 steps + mel VAE + HiFi-GAN), zeroscope (16 x 320x576 frames, 40 steps + VAE).
N > 1 (torchrun, one rank per GPU): every rank answers its own prompts; ONE gather of the padded outputs to rank 0.

    python scripts/bench_any2many.py [responses_per_rank=1]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_any2many.py
"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import FakeRobertaTokenizer, FakeTokenizer
from spider_amd import dp, routing
from spider_amd.clap import ClapTextConfig, ClapTextEngine
from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
from spider_amd.llm import LlamaEngine, LLMConfig
from spider_amd.pipelines import AudioLDMPipeline, StableDiffusionPipeline, TextToVideoSDPipeline
from spider_amd.schedulers import DDIMScheduler
from spider_amd.spider_decoder import SpiderDecoder
from spider_amd.unet import UNetConfig, UNetEngine
from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
from spider_amd.vae import VAEConfig, VAEDecoderEngine
from spider_amd.vocoder import HifiGanConfig, HifiGanEngine

n_resp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rank, world, local = dp.init_from_env()
dev = torch.device(f"cuda:{local}")
torch.cuda.set_device(dev)
P, T_NEW = 1536, 128
llm = LlamaEngine.random_init(LLMConfig.qwen25_7b(), dev, max_batch=1, max_len=P + T_NEW + 8, seed=0)
sd = StableDiffusionPipeline(UNetEngine.random_init(UNetConfig.sd15(), dev, 1), VAEDecoderEngine.random_init(VAEConfig.sd15(), dev, 2),
                             CLIPTextEngine.random_init(CLIPTextConfig.sd15(), dev, 3), FakeTokenizer(40000))
ad = AudioLDMPipeline(VAEDecoderEngine.random_init(VAEConfig.audioldm(), dev, 4), ClapTextEngine.random_init(ClapTextConfig(), dev, 5),
                      FakeRobertaTokenizer(40000), UNetEngine.random_init(UNetConfig.audioldm(), dev, 6),
                      DDIMScheduler(beta_start=0.0015, beta_end=0.0195), HifiGanEngine.random_init(HifiGanConfig.audioldm(), dev, 7))
vd = TextToVideoSDPipeline(UNet3DEngine.random_init(UNet3DConfig.zeroscope(), dev, 8), VAEDecoderEngine.random_init(VAEConfig.sd15(), dev, 9),
                           CLIPTextEngine.random_init(CLIPTextConfig(49408, 1024, 23, 16, 4096, 77, 1e-5, "gelu"), dev, 10), FakeTokenizer(40000))
decoder = SpiderDecoder(pipelines={"IMAGE": sd, "AUDIO": ad, "VIDEO": vd}, device=str(dev))
prompt = torch.randint(3, llm.cfg.vocab, (1, P), generator=torch.Generator(device=dev).manual_seed(2047 + rank), device=dev)
stage = {}


def respond():
    t0 = time.perf_counter()
    toks = llm.generate(input_ids=prompt, max_new_tokens=T_NEW, sync_every=T_NEW)[:, P:]
    head = " ".join(str(int(t)) for t in toks[0, :6].cpu())
    torch.cuda.synchronize(dev); t1 = time.perf_counter()
    # random-init weights emit no tags: the synthetic response carries exactly one caption per modality
    text = f"Sure. <IMAGE>scene {head}</IMAGE> <AUDIO>sound {head}</AUDIO> <VIDEO>clip {head}</VIDEO>"
    answers, preds, ptext = routing.new_outputs()
    answers, preds, ptext = decoder.generate({"llm_text_all": [text]}, answers, preds, ptext)
    torch.cuda.synchronize(dev); t2 = time.perf_counter()
    assert len(preds["IMAGE"]) == 1 and len(preds["AUDIO"]) == 1 and len(preds["VIDEO"]) == 1, {k: len(v) for k, v in preds.items()}
    img = torch.from_numpy(np.asarray(preds["IMAGE"][0], dtype=np.uint8))[None]            # [1, 512, 512, 3]
    aud = torch.from_numpy(np.asarray(preds["AUDIO"][0], dtype=np.float32).reshape(-1))[None]          # [1, 80000]
    vid = torch.from_numpy(np.stack([np.asarray(f, dtype=np.uint8) for f in preds["VIDEO"][0]]))[None]   # [1, 16, 320, 576, 3]
    stage.update(llm=t1 - t0, decoders=t2 - t1)
    return {"tokens": toks.to(torch.int32), "image": img.to(dev), "audio": aud.to(dev), "video": vid.to(dev)}


def step():
    return dp.gather_padded(respond(), 1, rank, world, dst=0)


step()
if world > 1:
    torch.distributed.barrier()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(n_resp):
    g = step()
torch.cuda.synchronize(dev)
if world > 1:
    torch.distributed.barrier()
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t)
if rank == 0:
    shapes = {k: list(v.shape) for k, v in g.items()}
    print(json.dumps({"metric": "any-to-many responses/sec (text -> text+image+audio+video)", "value": round(world * n_resp / dt, 4),
                      "n_gpus": world, "ms_per_response": round(dt / n_resp * 1e3, 1), "llm_ms": round(stage["llm"] * 1e3, 1),
                      "decoders_ms": round(stage["decoders"] * 1e3, 1), "gathered": shapes,
                      "gather_bytes_per_rank": int(sum(v[0].numel() * v[0].element_size() for k, v in g.items()))}))
if world > 1:
    torch.distributed.destroy_process_group()
