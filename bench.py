#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Spider any-to-many generation hot path on MI355X.

Metric (BASELINE.json): multimodal responses/sec (text -> text + image).
Workload (BASELINE.json configs[1]), synthetic, random-init weights of the true
shapes:
    one response = Qwen2.5-Omni-7B text-decoder shapes: prefill of a 1536-token prompt + 128 greedy tokens
                   -> signal-tag routing of a text carrying exactly one <IMAGE>caption</IMAGE>
                   -> CLIP-L/14 text encoder (cond + uncond) -> SD-v1.5 UNet, 64x64 latent (512^2 image),
                      PNDM 40 steps = 41 UNet calls at CFG batch 2, guidance 7.5 -> VAE decode to 512x512x3
The chain runs through the product class `spider_amd.SpiderFreeInfer` (the `predict` flow of qwen2.5omni_spider_web.py:458-521):
QwenOmniThinker.generate -> batch_decode -> extract_answer -> SpiderDecoderInfer -> SpiderDecoder.generate -> StableDiffusionPipeline;
bench.py builds random-init engines of the true shapes and a synthetic processor, hands requests to the class and packs the results.
A "step" is one such response per GPU (`--batch` prompts per GPU, default 1 = the reference's batch); `--schedule overlap` (default) is
the class's own pipelining of consecutive requests on two HIP streams (`SpiderFreeInfer.submit`).
N > 1: one process per GPU (torch.distributed / RCCL), prompts sharded with no data-path collective and ONE gather
of the padded outputs to rank 0 per step; weak scaling.

`python bench.py --gpus N` with no torchrun environment launches the N ranks itself (fresh child processes, decided before
any GPU call); under `python -m torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.
`--workload any2many` is BASELINE.json configs[4]: `--batch` (default 8) prompts per GPU, each answered with text + one
512^2 image (SD-v1.5) + 5 s of audio (AudioLDM-L) + a 16-frame 320x576 video (zeroscope) through SpiderDecoder.generate,
then the ONE gather of the padded outputs.

One JSON line on rank 0, with `roofline` (dominant kernel = the decode weight-streaming GEMV, HBM-bound, timed live with HIP events IN
THE CONDITION OF THE TIMED REGION -- beside the replaying decoder stream under `--schedule overlap` -- with the stand-alone figure next
to it) and `cpu_baseline` (the fp32 CPU oracle timed on this box's host cores on a bounded sample); at N = 1 also one timed step of the
any-to-many workload (`any2many`: configs[3]/[4], 8 prompts).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL between the ranks of one node needs dmabuf IPC on this driver stack (legacy IPC fails with `hipIpcGetMemHandle: invalid argument`);
# set before the HIP runtime loads, also when a launcher (torchrun) starts this file without the variable
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
MFMA_PEAK_TF = 2500.0  # dense bf16 / f16 MFMA peak
# 16-bit format of the diffusion engines (UNet / CLIP / VAE): f16 = the reference's torch_dtype (spider_decoder.py:109,114);
# bf16 on request. Same MFMA rate, same bytes: the two run at the same speed. The LLM is bf16.
DIFF_DT = torch.float16


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=None, help="prompts per GPU per step (default 1; 8 for --workload any2many)")
    p.add_argument("--workload", default="text_image", choices=["text_image", "any2many"],
                   help="text_image = BASELINE configs[1] (headline); any2many = configs[4] (text -> text+image+audio+video)")
    p.add_argument("--no-extras", action="store_true", help="skip the secondary timings (SDXL story / UNet3D / audio / Llama-8B)")
    p.add_argument("--no-any2many", action="store_true", help="skip the one-step any-to-many extra (configs[3]/[4], 8 prompts) of the N=1 line")
    p.add_argument("--prompt-len", type=int, default=1536)
    p.add_argument("--prompt-len-jitter", type=int, default=0,
                   help="synthetic prompts of different lengths (prompt_len - 0..J tokens, left-padded in a batch): exercises the "
                        "length-sorted sharding (dp.order_by_length) the data-parallel path uses; 0 = every prompt prompt_len tokens")
    p.add_argument("--new-tokens", type=int, default=128)
    p.add_argument("--denoise-steps", type=int, default=40)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--headline-only", action="store_true",
                   help="timed region + roofline objects only (no serial / batched / other-decoder timings, no CPU baseline): the command "
                        "the committed rocprofv3 summaries are taken from, so that their per-kernel averages are those of the timed region")
    p.add_argument("--schedule", default="overlap", choices=["overlap", "serial"],
                   help="text_image: overlap = response k's diffusion decoder on one stream beside response k+1's LLM pass on another "
                        "(one LLM pass + one decoder pass per step either way); serial = one stream")
    p.add_argument("--pipeline-depth", type=int, default=2, choices=[2, 3],
                   help="--schedule overlap: requests in flight in SpiderFreeInfer.submit: 2 = [LLM pass of k+1 | decoder pass of k]; "
                        "3 = [decode loop of k+1 | decoder pass of k, then the prompt pass of k+2] (the prompt pass rides on the decoder stream)")
    p.add_argument("--serial-decoders", action="store_true",
                   help="any2many: one SpiderDecoder.generate call per response (the reference's contract) instead of generate_batch")
    p.add_argument("--throughput-batch", type=int, default=8,
                   help="also report (outside the timed region, as an extra field) the rate with this many prompts per GPU; 0 = skip")
    p.add_argument("--llm", default="qwen25_7b", choices=["qwen25_7b", "llama3_8b"])
    p.add_argument("--precise", type=int, default=0, choices=[0, 1, 2],
                   help="precision level of the image decoder's UNet in the timed region (DESIGN.md section 4; 1 = the level whose full "
                        "41-evaluation loop latents sit inside north_star's 1e-3 for SD-v1.5). Default 0: the line's `precise_mode` "
                        "object carries the same schedule measured at level 1 beside the headline")
    p.add_argument("--no-stream32", action="store_true",
                   help="UNet residual stream in 16 bits instead of the fp32 master + 16-bit shadow the pipelines load by default")
    p.add_argument("--diffusion-dtype", default="f16", choices=["f16", "bf16"],
                   help="16-bit format of the UNet / text-encoder / VAE engines (f16 = the reference's torch_dtype)")
    a = p.parse_args()
    global DIFF_DT
    DIFF_DT = torch.float16 if a.diffusion_dtype == "f16" else torch.bfloat16
    if a.batch is None:
        a.batch = 8 if a.workload == "any2many" else 1
    return a


def launch_ranks(args, script=None, argv=None) -> int:
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (one per GPU, RCCL rendezvous on
    127.0.0.1) and return the worst exit code. Nothing in THIS process touches the GPU (device_count() does not
    initialise HIP); rank 0 prints the JSON line."""
    import socket
    import subprocess
    n_vis = torch.cuda.device_count()
    if n_vis < args.gpus and not os.environ.get("SPIDER_SHARE_GPU"):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_vis} GPU(s) visible on this node")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv), env=env))
    rc = 0
    try:
        for pr in procs:
            rc = max(rc, abs(pr.wait()))
            if rc:          # one rank failed: the others would wait in a collective forever
                break
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def _masked_streams(dev):
    """Tuning aid: CU-masked HIP streams for the two passes of the pipelined schedule (hipExtStreamCreateWithCUMask).
    SPIDER_BENCH_CUMASK_L / SPIDER_BENCH_CUMASK_U = "<n>[:<first>]": n consecutive CUs starting at `first` for the LLM / decoder stream."""
    specs = os.environ.get("SPIDER_BENCH_CUMASK_L"), os.environ.get("SPIDER_BENCH_CUMASK_U")
    if not any(specs):
        return None
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def make(spec, prio):
        if not spec:
            return torch.cuda.Stream(device=dev, priority=prio)
        n, _, first = spec.partition(":")
        n, first = int(n), int(first or 0)
        bits = [0] * 8
        for i in range(first, min(256, first + n)):
            bits[i // 32] |= 1 << (i % 32)
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, (ctypes.c_uint32 * 8)(*bits))
        assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
        return torch.cuda.ExternalStream(st.value, device=dev)
    return make(specs[0], 0), make(specs[1], -1)


class _PlainThinker:
    """`--llm llama3_8b`: a text decoder without multimodal rotary sections behind the thinker's `generate` surface."""

    def __init__(self, llm):
        self.llm = llm

    def generate(self, input_ids, attention_mask=None, **kw):
        kw.pop("spk", None); kw.pop("use_audio_in_video", None)
        return self.llm.generate(input_ids=input_ids, attention_mask=attention_mask, **kw)


class Responder:
    """One GPU's responder: a thin caller of the product class `spider_amd.SpiderFreeInfer` (the `predict` flow of
    qwen2.5omni_spider_web.py:458-521): QwenOmniThinker.generate -> batch_decode -> extract_answer -> SpiderDecoderInfer ->
    SpiderDecoder.generate -> StableDiffusionPipeline (tokenizer, CLIP text encoder, 41 UNet evaluations, VAE, PIL). Nothing of the
    hot path lives here: bench.py builds random-init engines of the true shapes, a synthetic processor (no vocabulary files exist;
    random weights emit no signal tags, so its batch_decode renders one <IMAGE> caption per response), hands requests to the class
    and packs the results into fixed-shape tensors for the gather. `--schedule overlap` = the class's own two-stream pipelining of
    consecutive requests (`SpiderFreeInfer.submit`)."""
    TAGS = ("IMAGE",)

    def __init__(self, args, device, rank=0):
        from spider_amd import SpiderDecoderInfer, SpiderFreeInfer
        from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
        from spider_amd.llm import LlamaEngine, LLMConfig
        from spider_amd.pipelines import StableDiffusionPipeline
        from spider_amd.qwen_omni import QwenOmniThinker
        from benchkit.synthetic import FakeTokenizer, SyntheticOmniProcessor
        from spider_amd.unet import UNetConfig, UNetEngine
        from spider_amd.vae import VAEConfig, VAEDecoderEngine
        self.args, self.dev = args, device
        a, dev, D = args, device, DIFF_DT
        cfg = getattr(LLMConfig, a.llm)()
        self.max_batch = max(a.batch, min(a.throughput_batch, 8)) if self.TAGS == ("IMAGE",) else min(a.batch, 8)
        self.llm = LlamaEngine.random_init(cfg, dev, max_batch=self.max_batch, max_len=a.prompt_len + a.new_tokens + 8, seed=0)
        self.unet = UNetEngine.random_init(UNetConfig.sd15(), dev, seed=1, dtype=D, stream32=not a.no_stream32, precise=getattr(a, "precise", 0))
        self.sd = StableDiffusionPipeline(self.unet, VAEDecoderEngine.random_init(VAEConfig.sd15(), dev, 3, dtype=D),
                                          CLIPTextEngine.random_init(CLIPTextConfig.sd15(), dev, 2, dtype=D), FakeTokenizer(40000))
        self.sched = self.sd.scheduler
        pipes = {"IMAGE": self.sd, **self.more_pipelines(dev, D)}
        steps = {m: {"num_inference_steps": a.denoise_steps} for m in pipes}
        self.decoder_infer = SpiderDecoderInfer({"model": dict(type="spider_decoder", pipelines=pipes, device=str(dev), decode_kwargs=steps)})
        self.decoder = self.decoder_infer.spider_decoder
        thinker = QwenOmniThinker(self.llm) if cfg.mrope_section else _PlainThinker(self.llm)
        self.processor = SyntheticOmniProcessor(cfg.vocab, tags=self.TAGS, prompt_len=a.prompt_len)
        # eos_token_id=[]: random-init weights would emit an EOS id at a random step; every response carries exactly new_tokens tokens
        depth = getattr(a, "pipeline_depth", 2) if hasattr(thinker, "prefill_begin") else 2
        self.infer = SpiderFreeInfer(thinker, self.processor, self.decoder_infer, device=dev, depth=depth,
                                     generate_kwargs=dict(max_new_tokens=a.new_tokens, eos_token_id=[], sync_every=a.new_tokens))
        self.infer._streams = _masked_streams(dev)       # tuning aid (SPIDER_BENCH_CUMASK_L / _U); None = the class's own two streams
        g = torch.Generator(device=dev).manual_seed(2047 + rank)  # seed echoes Comic_Generation.py:387
        # This rank's prompts of the global request list: the world x rows synthetic prompts are sorted by expected
        # length (dp.order_by_length: variable token counts are the main load-imbalance source), then dealt out strided, prompt i to
        # rank i mod world; every prompt has a seed of its own (its global index), shorter ones are LEFT-padded like a processor batch.
        from spider_amd import dp
        world, mb, J = max(1, getattr(a, "gpus", 1)), self.max_batch, max(0, getattr(a, "prompt_len_jitter", 0))
        lengths = [a.prompt_len - ((i * 2654435761) >> 7) % (J + 1) for i in range(world * mb)]
        order = dp.order_by_length(lengths)
        self.prompt_ids = [order[i] for i in dp.shard_indices(world * mb, rank, world)]
        self.prompt_lens = [lengths[i] for i in self.prompt_ids]
        self.prompt = torch.zeros(mb, a.prompt_len, dtype=torch.long, device=dev)
        self.prompt_mask = torch.zeros(mb, a.prompt_len, dtype=torch.long, device=dev)
        for b, (gid, n) in enumerate(zip(self.prompt_ids, self.prompt_lens)):
            gg = torch.Generator(device=dev).manual_seed(2047 + gid)
            self.prompt[b, a.prompt_len - n:] = torch.randint(3, cfg.vocab, (n,), generator=gg, device=dev)
            self.prompt_mask[b, a.prompt_len - n:] = 1
        self.latents0 = torch.randn(self.max_batch, 4, 64, 64, generator=g, device=dev)
        self.enc_synth = torch.randn(2 * self.max_batch, 77, 768, generator=g, device=dev).to(D)
        self.overlap_ms, self.stage = None, {}
        self.step_log = []          # one record per respond(): device span of each pass on its stream + host wall time of its pieces
        self._B = None

    def more_pipelines(self, dev, D):
        return {}

    @property
    def _streams(self):
        return self.infer._streams

    def includes(self):
        return ["chat_inputs", "llm_prefill", "llm_decode", "batch_decode", "routing", "clip_tokenize", "clip_text_encoder", "unet_denoise_loop",
                "vae_decode", "pil_conversion"]

    def request(self, B):
        # what the processor emits for a (left-padded) batch of text-only chats
        return {"input_ids": self.prompt[:B].contiguous(), "attention_mask": self.prompt_mask[:B].contiguous()}

    def pack(self, res, B):
        """results of one request (B rows) -> fixed-shape device tensors for the gather"""
        import numpy as np
        res = res if isinstance(res, list) else [res]
        a = self.args
        toks = torch.stack([r.text_ids[a.prompt_len:] for r in res]).to(torch.int32)
        out = {"tokens": toks.to(self.dev)}
        for r in res:
            assert all(len(r.predictions[m]) == 1 for m in self.TAGS), "every response carries one output per requested modality"
        out["out" if self.TAGS == ("IMAGE",) else "image"] = torch.from_numpy(np.stack([np.asarray(r.predictions["IMAGE"][0], dtype=np.uint8) for r in res])).permute(0, 3, 1, 2).contiguous().to(self.dev)
        if "AUDIO" in self.TAGS:
            out["audio"] = torch.from_numpy(np.stack([np.asarray(r.predictions["AUDIO"][0], dtype=np.float32).reshape(-1) for r in res])).to(self.dev)
        if "VIDEO" in self.TAGS:
            out["video"] = torch.from_numpy(np.stack([np.stack([np.asarray(f, dtype=np.uint8) for f in r.predictions["VIDEO"][0]]) for r in res])).to(self.dev)
        return out

    def respond_serial(self, batch=None):
        """One request start to finish on one stream (the latency of ONE request; the reference's schedule)."""
        B = batch or self.args.batch
        while self.infer.flush() is not None:       # requests still in flight under the pipelined schedule: drain them first
            pass
        self.decoder.stage_events = {}
        out = self.pack(self.infer.predict(inputs=self.request(B)), B)
        self.stage = self.decoder.stage_ms_device()
        self.step_log.append({**(self.infer.last_pass_ms or {}), **self.infer.host_ms})
        return out

    def respond(self, batch=None):
        """One step = one LLM pass + one decoder pass (`--schedule overlap`, default: `SpiderFreeInfer.submit` -- the decoder pass of
        request k on one HIP stream beside the LLM pass of request k+1 on another; the step returns the finished request k.
        `--schedule serial`: `SpiderFreeInfer.predict`, the two passes of the same request back to back on one stream)."""
        B = batch or self.args.batch
        if self.args.schedule == "serial":
            return self.respond_serial(B)
        if self._B != B:                            # another request geometry: drain the pipeline first
            while self.infer.flush() is not None:
                pass
            self._B = B
        self.decoder.stage_events = {}
        res = self.infer.submit(inputs=self.request(B))
        while res is None:                          # empty pipeline (first call): prime it (2 or 3 submits); untimed warm-up work
            res = self.infer.submit(inputs=self.request(B))
        self.overlap_ms = self.infer.last_pass_ms or None
        self.stage = self.decoder.stage_ms_device()
        ms = torch.cuda.memory_stats(self.dev)       # segments the caching allocator took from / gave back to the driver so far
        self.step_log.append({**(self.infer.last_pass_ms or {}), **self.infer.host_ms,
                              "device_allocs": ms.get("num_device_alloc", 0), "device_frees": ms.get("num_device_free", 0)})
        return self.pack(res, B)


class AnyToManyResponder(Responder):
    """BASELINE configs[3]/[4]: text -> {text + image + audio + video}: the same product path with a response that carries one
    caption per modality; the rank's `batch` prompts are answered by ONE batched generate and ONE SpiderDecoder.generate_batch
    (`--serial-decoders`: one SpiderDecoder.generate per response, the reference's contract, spider_decoder.py:311). Random-init
    weights of the true shapes: SD-v1.5, AudioLDM-L (train_configs/spider_decoder_cfg.py:37), zeroscope_v2_576w."""
    TAGS = ("IMAGE", "AUDIO", "VIDEO")

    def more_pipelines(self, dev, D):
        from spider_amd.clap import ClapTextConfig, ClapTextEngine
        from spider_amd.clip import CLIPTextConfig, CLIPTextEngine
        from spider_amd.pipelines import AudioLDMPipeline, TextToVideoSDPipeline
        from spider_amd.schedulers import DDIMScheduler
        from benchkit.synthetic import FakeRobertaTokenizer, FakeTokenizer
        from spider_amd.unet import UNetConfig, UNetEngine
        from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
        from spider_amd.vae import VAEConfig, VAEDecoderEngine
        from spider_amd.vocoder import HifiGanConfig, HifiGanEngine
        s32 = not self.args.no_stream32
        ad = AudioLDMPipeline(VAEDecoderEngine.random_init(VAEConfig.audioldm(), dev, 4, dtype=D), ClapTextEngine.random_init(ClapTextConfig(), dev, 5, dtype=D),
                              FakeRobertaTokenizer(40000), UNetEngine.random_init(UNetConfig.audioldm_l(), dev, 6, dtype=D, stream32=s32),
                              DDIMScheduler(beta_start=0.0015, beta_end=0.0195), HifiGanEngine.random_init(HifiGanConfig.audioldm(), dev, 7, dtype=D))
        vd = TextToVideoSDPipeline(UNet3DEngine.random_init(UNet3DConfig.zeroscope(), dev, 8, dtype=D, stream32=s32), VAEDecoderEngine.random_init(VAEConfig.sd15(), dev, 9, dtype=D),
                                   CLIPTextEngine.random_init(CLIPTextConfig(49408, 1024, 23, 16, 4096, 77, 1e-5, "gelu"), dev, 10, dtype=D),
                                   FakeTokenizer(40000))
        return {"AUDIO": ad, "VIDEO": vd}

    def includes(self):
        return ["chat_inputs", "llm_prefill", "llm_decode", "batch_decode", "routing", "clip_text_encoder", "sd15_unet_loop", "vae_decode",
                "clap_text_encoder", "audioldm_l_unet_loop", "mel_vae_decode", "hifigan_vocoder", "zeroscope_unet3d_loop", "vae_decode_16_frames"]

    def respond_serial(self, batch=None):
        if not self.args.serial_decoders:
            return super().respond_serial(batch)
        B = batch or self.args.batch                # one SpiderFreeInfer.predict per response: batch 1 everywhere
        outs = [self.pack(self.infer.predict(inputs={k: v[b:b + 1] for k, v in self.request(B).items()}), 1) for b in range(B)]
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}

    def respond(self, batch=None):
        if self.args.serial_decoders:
            return self.respond_serial(batch)
        return super().respond(batch)


def _unet_input(unet, lat):
    """CFG-batch-2 NHWC input of one UNet evaluation in the form the engine's mode reads (precise: the un-rounded fp32 latents)"""
    from spider_amd import ops
    return ops.latent_to_nhwc_f32(lat, reps=2) if getattr(unet, "precise", 0) else ops.latent_to_nhwc(lat, reps=2, dtype=DIFF_DT)


def precise_mode_leg(args, resp, device, level=1, steps=5):
    """The headline schedule once more with the image decoder's UNet at `precise=level` -- the mode whose latents after the full
    41-evaluation loop sit inside north_star's 1e-3 of the fp32 oracle (tests/test_fullsize_parity.py) -- so the line says what that
    bound costs in the headline's own unit. Same responder, same requests, same two-stream schedule; only `pipeline.unet` differs."""
    from spider_amd.unet import UNetConfig, UNetEngine
    pu = UNetEngine.random_init(UNetConfig.sd15(), device, seed=1, dtype=DIFF_DT, stream32=True, precise=level)
    keep = resp.sd.unet
    while resp.infer.flush() is not None:
        pass
    resp._B = None
    resp.sd.unet = pu
    resp.infer.reset_warm()                      # the new engine has no graphs: its first pass runs alone on the calling thread
    try:
        for _ in range(3):                       # serial capture pass, pipeline primed, one overlapped step
            resp.respond()
        torch.cuda.synchronize(device)
        n0, walls = len(resp.step_log), []
        for _ in range(steps):
            t1 = time.perf_counter()
            resp.respond()
            walls.append((time.perf_counter() - t1) * 1e3)
        torch.cuda.synchronize(device)
        log = resp.step_log[n0:]
        del resp.step_log[n0:]
        ts = resp.sched.set_timesteps(args.denoise_steps)
        x2 = _unet_input(pu, resp.latents0[:1].contiguous())
        pu.prepare(ts, resp.enc_synth[:2].contiguous())
        pu.step(x2, 0)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10):
            pu.step(x2, i % len(ts))
        e1.record(); e1.synchronize()
        med = sorted(walls)[len(walls) // 2]
        return {"precise": level, "responses_per_s": round(args.batch / (med * 1e-3), 4), "ms_per_step": _dist4(walls),
                "llm_pass_ms": _dist4([r["llm_pass_ms"] for r in log if "llm_pass_ms" in r])["median"] if log else None,
                "decoder_pass_ms": _dist4([r["decoder_pass_ms"] for r in log if "decoder_pass_ms" in r])["median"] if log else None,
                "unet_step_ms": round(e0.elapsed_time(e1) / 10, 3), "schedule": args.schedule, "steps_timed": steps,
                "latents_vs_fp32_oracle": "SD-v1.5 [2,4,64,64], 41 evaluations, CFG 7.5: relative L2 5.7e-4 at this level, 1.26e-3 at level 0 "
                                          "(tests/test_fullsize_parity.py::test_sd15_full_41_step_loop_latents_precise_mode_inside_1e3)"}
    finally:
        while resp.infer.flush() is not None:
            pass
        resp._B = None
        resp.sd.unet = keep
        resp.infer.reset_warm()


def measure_roofline(resp, device):
    """Dominant kernel: gemv_kernel<1,1,GATEUP> (fused gate/up projection + SwiGLU of one decoded token), the largest weight stream of
    the decode step. Algorithmic bytes per launch = 2*I*H*2 (weights) + H*2 + I*2. Timed live with HIP events around back-to-back
    launches on the stream they are launched on, cycling through all layers' weights.
    `achieved` / `frac` are taken IN THE CONDITION OF THE TIMED REGION: under `--schedule overlap` the loop runs on the LLM stream while
    UNet evaluations replay on the decoder stream, exactly as the two passes of a step share the chip (the kernel runs longer there and
    its launches start later: profiles/r0N_bench_kernel_stats_top.txt holds rocprofv3's average for the same kernel in the same
    condition); `standalone` is the same loop with the chip to itself. Under `--schedule serial` the two are the same measurement."""
    from spider_amd import ops
    llm = resp.llm
    c = llm.cfg
    B = 1
    x = torch.randn(B, c.hidden, device=device).to(torch.bfloat16)
    out = torch.empty(B, c.inter, dtype=torch.bfloat16, device=device)
    layers = llm.layers
    for lw in layers[:4]:
        ops.gemv_swiglu(lw["w_gu"], x, norm_w=lw["ln2"], eps=c.eps, out=out)
    torch.cuda.synchronize(device)
    reps = 4
    n = reps * len(layers)

    def timed_loop(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(reps):
                for lw in layers:   # cycle through all layers' weights: 28 x 272 MB >> 256 MiB Infinity Cache
                    ops.gemv_swiglu(lw["w_gu"], x, norm_w=lw["ln2"], eps=c.eps, out=out)
            e1.record(stream)
        return e0, e1

    cur = torch.cuda.current_stream(device)
    e0, e1 = timed_loop(cur)
    e1.synchronize()
    us_alone = e0.elapsed_time(e1) * 1e3 / n
    us = us_alone
    corun = resp.args.schedule == "overlap" and resp._streams is not None
    if corun:
        sL, sU = resp._streams
        x2 = _unet_input(resp.unet, resp.latents0[:1].contiguous())
        sU.wait_stream(cur); sL.wait_stream(cur)
        with torch.cuda.stream(sU), ops.workspace_scope("image"):
            for i in range(6):                       # >= 30 ms of UNet evaluations: covers the ~6 ms GEMV loop below
                resp.unet.step(x2, i)
        e0, e1 = timed_loop(sL)
        e1.synchronize()
        sU.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
    bytes_alg = 2 * c.inter * c.hidden * 2 + c.hidden * 2 + c.hidden * 2 + c.inter * 2
    achieved = bytes_alg / (us * 1e-6) / 1e9
    alone = bytes_alg / (us_alone * 1e-6) / 1e9
    # HBM bytes per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction); only
    # valid for the shapes it was collected on
    traffic, src = None, None
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        pm = os.path.join(ROOT, "profiles", f"{tag}_pmc_decode_hbm.json")
        if os.path.exists(pm) and bytes_alg == 271633408:
            try:
                traffic = json.load(open(pm))["dominant_kernel"]["hbm_read_bytes_corrected"]
                src = f"profiles/{tag}_pmc_decode_hbm.json"
                break
            except Exception:
                pass
    return {"bound": "hbm", "kernel": "gemv_kernel<NB=1,R=1,GATEUP=1,XLDS=1> (decode gate/up + SwiGLU)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": src, "avg_launch_us": round(us, 2),
            "algorithmic_bytes_per_launch": bytes_alg, "launches_timed": n,
            "condition": ("two-stream schedule of the timed region: loop on the LLM stream while UNet evaluations replay on the decoder "
                          "stream; avg_launch_us is the launch-to-launch interval of back-to-back launches (kernel + dispatch gap)"
                          if corun else "one stream (the schedule of the timed region)"),
            "standalone": {"avg_launch_us": round(us_alone, 2), "achieved": round(alone, 1), "frac": round(alone / HBM_PEAK_GBS, 4)}}


def _unet_pmc_traffic(kernel_prefix):
    """HBM bytes per launch (read + write) of a UNet kernel from the committed counter pass (profiles/r0N_pmc_unet_hbm.json:
    averages over every launch of that kernel in the UNet step, not only the roofline shape -- stated in `traffic_source`)."""
    for tag in ("r06", "r05", "r04", "r03", "r02"):          # newest committed counter pass first
        pm = os.path.join(ROOT, "profiles", f"{tag}_pmc_unet_hbm.json")
        try:
            for k in json.load(open(pm))["kernels"]:
                if k["kernel"].startswith(kernel_prefix):
                    return (k["hbm_read_bytes"] + k["hbm_write_bytes"],
                            f"profiles/{tag}_pmc_unet_hbm.json (average over the {k['launches']} launches of this kernel in the profiled UNet steps)")
        except Exception:
            pass
    return None, None


def measure_mfma_roofline(device):
    """Largest MFMA-bound kernel class of the UNet step: the 3x3 implicit-GEMM conv (ResnetBlock2D conv at the 64x64 latent,
    320 -> 320 channels, CFG batch 2: M = 8192 output pixels, N = 320, K = 2880; the LDS-DMA kernel on 64-row tiles, no split-K).
    Algorithmic flops per call = 2*M*N*K; timed live with HIP events over back-to-back calls on torch's current stream."""
    from spider_amd import ops
    x = torch.randn(2, 64, 64, 320, device=device).to(DIFF_DT)
    w = (torch.randn(320, 3, 3, 320, device=device) * 0.02).to(DIFF_DT)
    # gn_groups=32: the instantiation the UNet step runs (ResnetBlock2D conv1 / conv2 leave the GroupNorm partials of their output
    # in the epilogue, round 4)
    for _ in range(5):
        ops.conv2d(x, w, gn_groups=32)
    torch.cuda.synchronize(device)
    stream = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    g = torch.cuda.CUDAGraph()      # graph replay: the ~10 us of Python/ctypes launch overhead would otherwise be in the figure
    with torch.cuda.graph(g):
        for _ in range(n):
            ops.conv2d(x, w, gn_groups=32)
    g.replay()
    torch.cuda.synchronize(device)
    e0.record(stream)
    g.replay()
    e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = 2 * 8192 * 320 * 2880
    tf = flops / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "gemm_dma_kernel<160,4,CONV,BM=64,GN> (UNet 3x3 conv, 64x64 latent, 320->320, batch 2; 256 tiles of 64 x 160, no split-K; "
                                       "GroupNorm partial statistics of the output written by the epilogue)",
            "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
            "traffic": _unet_pmc_traffic("gemm_dma_kernel<160, 4, true, 0, 64, true>")[0], "traffic_source": _unet_pmc_traffic("gemm_dma_kernel<160, 4, true, 0, 64, true>")[1],
            "avg_call_us": round(us, 2), "algorithmic_flops_per_call": flops, "calls_timed": n}


def measure_prefill_gemm_roofline(device, cfg, prompt_len):
    """Largest MFMA-bound kernel of the LLM prefill: the fused gate/up projection (LlamaMLP.gate_proj / up_proj over the prompt,
    modeling_llama3.py:197-199): [prompt, 2 * inter, hidden] on the 256 x 256 LDS-DMA kernel. Timed live, graph-replayed."""
    from spider_amd import ops
    M, N, K = prompt_len, 2 * cfg.inter, cfg.hidden
    A = torch.randn(M, K, device=device).to(torch.bfloat16)
    W = (torch.randn(N, K, device=device) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(A, W, out=out)
    torch.cuda.synchronize(device)
    n = 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            ops.gemm(A, W, out=out)
    g.replay()
    torch.cuda.synchronize(device)
    stream = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    g.replay()
    e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = 2 * M * N * K
    tf = flops / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": f"gemm_p8_kernel (LLM prefill gate/up projection, {M} x {N} x {K}, 256 x 256 tiles)",
            "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4), "traffic": None,
            "avg_call_us": round(us, 2), "algorithmic_flops_per_call": flops, "calls_timed": n}


def measure_attention_roofline(device):
    """The UNet's largest attention: self-attention at the 64x64 latent (4096 tokens, 8 heads, d = 40, CFG batch 2).
    Algorithmic flops = 4 * N^2 * C * B (QK^T + PV, d = 40 as stored -- the kernel pads d to 64 inside its tiles)."""
    from spider_amd import ops
    N, heads, d, B = 4096, 8, 40, 2
    C = heads * d
    qkv = torch.randn(B, N, 3 * C, device=device).to(DIFF_DT)
    f = lambda: ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
    for _ in range(3):
        f()
    torch.cuda.synchronize(device)
    n = 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            f()
    g.replay()
    torch.cuda.synchronize(device)
    stream = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    g.replay()
    e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = 4 * N * N * C * B
    tf = flops / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "attn_flash_pipe_kernel<64> (UNet self-attention, 64x64 latent: 4096 tokens, 8 heads, d=40, batch 2)",
            "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
            "traffic": _unet_pmc_traffic("attn_flash_pipe_kernel<64, true,")[0], "traffic_source": _unet_pmc_traffic("attn_flash_pipe_kernel<64, true,")[1],
            "avg_call_us": round(us, 2), "algorithmic_flops_per_call": flops, "calls_timed": n}


def cpu_baseline(args):
    """fp32 CPU oracle ("port") on a bounded sample, extrapolated linearly to one response."""
    from oracle.llama import LlamaCfg, LlamaOracle
    from oracle.unet import PNDMOracle, UNetCfg, UNetOracle, random_unet_weights
    # 32 threads: the oracle's fp32 torch ops stop scaling (and regress badly from oversubscription) beyond that on
    # the GPU box's many-core host; `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    full = getattr(LlamaCfg, args.llm)()
    c1 = LlamaCfg(**{**full.__dict__, "layers": 1, "tie_embeddings": True})
    w = LlamaOracle.random_weights(c1, seed=0, std=0.02, bf16_round=False)
    m = LlamaOracle(c1, w)
    S = args.prompt_len
    ids = torch.randint(3, c1.vocab, (1, S))
    pos = torch.arange(S)[None]
    with torch.no_grad():
        t0 = time.perf_counter()
        h = m.w["model.embed_tokens.weight"][ids]
        # one decoder layer over the whole prompt (forward() also runs the lm_head over all positions; time the layer
        # alone by calling with a 1-token lm_head: slice afterwards is what HF does with logits_to_keep)
        logits, kv, _ = m.forward(ids[:, :S], pos, None, None)
        t_prefill_layer_plus_head = time.perf_counter() - t0
        t0 = time.perf_counter()
        torch.nn.functional.linear(torch.randn(S, c1.hidden), m.w["lm_head.weight"])
        t_head_S = time.perf_counter() - t0
        t_prefill_layer = max(t_prefill_layer_plus_head - t_head_S, 1e-6)
        nd = 3
        t0 = time.perf_counter()
        for i in range(nd):
            logits, kv, _ = m.forward(torch.randint(3, c1.vocab, (1, 1)), torch.tensor([[S + i]]), kv, None)
        t_dec = (time.perf_counter() - t0) / nd
        t0 = time.perf_counter()
        torch.nn.functional.linear(torch.randn(1, c1.hidden), m.w["lm_head.weight"])
        t_head1 = time.perf_counter() - t0
    t_dec_layer = max(t_dec - t_head1, 1e-6)
    del m, w
    ucfg = UNetCfg.sd15()
    uo = UNetOracle(ucfg, random_unet_weights(ucfg, seed=0, bf16_round=False))
    x = torch.randn(2, 4, 64, 64)
    enc = torch.randn(2, 77, 768)
    uo.forward(x[:, :, :8, :8], torch.tensor(10), enc)  # warm
    t0 = time.perf_counter()
    uo.forward(x, torch.tensor(500), enc)
    t_unet = time.perf_counter() - t0
    L = full.layers
    n_unet = args.denoise_steps + 1
    t_resp = L * t_prefill_layer + t_head1 + args.new_tokens * (L * t_dec_layer + t_head1) + n_unet * t_unet
    return {"value": round(1.0 / t_resp, 6), "unit": "responses/s", "cores": cores, "kind": "port", "extrapolated": True,
            "sample": (f"oracle fp32 on {cores} host threads: 1 of {L} decoder layers over the {S}-token prompt "
                       f"({t_prefill_layer:.2f}s) + {nd} decode tokens through 1 layer at context {S} ({t_dec_layer * 1e3:.1f} ms/layer-token) "
                       f"+ lm_head ({t_head1 * 1e3:.0f} ms) + 1 SD-v1.5 UNet step at [2,4,64,64] ({t_unet:.2f}s); extrapolated "
                       f"linearly to {L} layers x ({S} prompt + {args.new_tokens} new tokens) + {n_unet} UNet calls "
                       "(text encoder / VAE not included in the CPU figure)"),
            "unet_step_ms": round(t_unet * 1e3, 1), "decode_tokens_per_s": round(1.0 / (L * t_dec_layer + t_head1), 4)}


def measure_story_attention_roofline(device):
    """Consistent self-attention of the StoryDiffusion write phase at 768^2 (Comic_Generation.py:94-118): the [8, N, C] ->
    [2, 4N, C] view at the 48x48 up-block level: 9216 tokens, 10 heads, d = 64, column-structured keep mask (sa64 = 0.5), run
    as the pipeline runs it: visible-key lists built once per step, masked keys skipped. `achieved` counts the flops EXECUTED
    (4 * N * visible keys per image * C * 2 groups); the dense count the reference computes (every key scored, masked ones zeroed;
    SURVEY.md section 8d) is reported beside it."""
    from spider_amd import ops
    from spider_amd.story import pack_keep_bits
    N, img, heads, C = 2304, 4, 10, 640
    L = img * N
    qkv = torch.randn(2, L, 3 * C, device=device).to(DIFF_DT)
    keep = torch.rand(L, generator=torch.Generator().manual_seed(0)) < 0.5
    bits = pack_keep_bits(keep).to(device)
    ki, tl = ops.story_key_lists(bits, L, N, 0, img, 0)
    f = lambda: ops.attention_keylist(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, ki, tl)
    for _ in range(2):
        f()
    torch.cuda.synchronize(device)
    n = 5
    stream = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n):
        f()
    e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tpl = (N + 127) // 128
    visible = int(tl.view(img, tpl, 4)[:, 0, 3].sum().item())          # sum over the 4 images of their visible keys
    flops = 4 * N * visible * C * 2
    dense = 4 * L * L * C * 2
    tf = flops / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "attn_flash_kernel<64, KIDX> (SDXL consistent self-attention through visible-key lists, 768^2: "
                                       "9216 tokens, 10 heads, d=64, 2 CFG groups)",
            "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4), "traffic": None,
            "avg_call_us": round(us, 2), "algorithmic_flops_per_call": flops, "calls_timed": n,
            "visible_keys_per_image": visible // img, "dense_flops_per_call": dense,
            "dense_equivalent_tflops": round(dense / (us * 1e-6) / 1e12, 1),
            "note": "executed flop count: keys the keep vector masks for an image are never loaded or scored (they contribute exactly "
                    "zero in the reference, which scores all 9216 and zeroes them); dense_equivalent_tflops prices the same call at "
                    "the reference's dense count"}


def _ev_ms(fn, n, device):
    stream = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record(stream)
    for i in range(n):
        fn(i)
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def other_decoders(device):
    """Secondary, driver-visible numbers for the other BASELINE configs (outside the timed region; random-init weights of the
    true shapes): SDXL story step (configs[2]), UNet3D step + AudioLDM-L clip (configs[3]/[4]), Llama-8B decode (configs[0]/[2])."""
    import random as _random
    from spider_amd import ops
    from spider_amd.schedulers import DDIMScheduler
    from spider_amd.unet import UNetConfig, UNetEngine, unet_flops
    out = {}
    g = torch.Generator(device=device).manual_seed(7)
    # ---- SDXL story step: 768^2, CFG batch 8 (4 panels), FreeU, all 36 up-block processors on the consistent path
    from spider_amd.story import ConsistentSelfAttention, StoryState
    sdxl = UNetEngine.random_init(UNetConfig.sdxl(), device, seed=11, dtype=DIFF_DT)
    sdxl.freeu = (0.6, 0.4, 1.1, 1.2)
    hw = 96
    x = torch.randn(8, hw, hw, 4, generator=g, device=device).to(DIFF_DT)
    enc = torch.randn(8, 77, 2048, generator=g, device=device).to(DIFF_DT)
    added = dict(text_embeds=torch.randn(8, 1280, generator=g, device=device).to(DIFF_DT),
                 time_ids=torch.tensor([[768, 768, 0, 0, 768, 768]] * 8, dtype=torch.float32))
    ts = DDIMScheduler().set_timesteps(50)
    sdxl.prepare(ts, enc, added)
    rnd = _random.Random(2047)
    st = StoryState(total_count=ConsistentSelfAttention.count_processors(sdxl), height=768, width=768, id_length=4, sa32=0.5, sa64=0.5,
                    write=True, cur_step=5, coin=rnd.random)
    st.regen_masks(device)
    sdxl.self_attn_hook = ConsistentSelfAttention(st)
    sdxl.step(x, 5)
    n = 6
    ms = _ev_ms(lambda i: sdxl.step(x, 6 + i), n, device)
    sdxl.self_attn_hook = None
    fl = unet_flops(UNetConfig.sdxl(), hw, hw)
    plain_up = 6 * 4 * 2304 ** 2 * 640 + 30 * 4 * 576 ** 2 * 1280            # the 36 up-block self-attentions, per sample
    cons = 2 * (6 * 4 * 9216 ** 2 * 640 + 30 * 4 * 2304 ** 2 * 1280)         # the same layers on the consistent path, per step
    p_cons = 0.7                                                             # Bernoulli(0.7) for steps 5..19 (Comic_Generation.py:94-118)
    step_flops = 8 * fl["total"] + p_cons * (cons - 8 * plain_up)
    out["sdxl_story_step_ms"] = round(ms, 2)
    out["sdxl_story_step"] = {"latent": "8x96x96x4 (4 panels, CFG), 768^2", "consistent_self_attention": "36 up-block processors, coin 0.7 (steps 5..19)",
                              "freeu": True, "expected_flops_per_step_tf": round(step_flops / 1e12, 2),
                              "tflops_per_s": round(step_flops / (ms * 1e-3) / 1e12, 1), "mfma_frac": round(step_flops / (ms * 1e-3) / 1e12 / 2500.0, 4),
                              "flops_per_sample_plain_gf": {k: round(v / 1e9, 1) for k, v in fl.items()}}
    out["roofline_story_attention"] = measure_story_attention_roofline(device)
    del sdxl
    torch.cuda.empty_cache()
    # ---- zeroscope UNet3D step: CFG batch 2 x 16 frames at 40x72
    from spider_amd.unet3d import UNet3DConfig, UNet3DEngine
    u3 = UNet3DEngine.random_init(UNet3DConfig.zeroscope(), device, seed=12, dtype=DIFF_DT, stream32=True)
    enc = torch.randn(2, 77, 1024, generator=g, device=device).to(DIFF_DT)
    ts = DDIMScheduler().set_timesteps(40)
    u3.prepare(ts, enc, frames=16)
    x2 = ops.latent_to_nhwc(torch.randn(16, 4, 40, 72, generator=g, device=device), reps=2, dtype=DIFF_DT)
    u3.step(x2, 0)
    out["unet3d_step_ms"] = round(_ev_ms(lambda i: u3.step(x2, i), 5, device), 2)
    del u3
    torch.cuda.empty_cache()
    # ---- AudioLDM-L clip: 5 s, 40 DDIM steps, CFG, mel VAE, HiFi-GAN (train_configs/spider_decoder_cfg.py:37)
    from spider_amd.clap import ClapTextConfig, ClapTextEngine
    from spider_amd.pipelines import AudioLDMPipeline
    from spider_amd.vae import VAEConfig, VAEDecoderEngine
    from spider_amd.vocoder import HifiGanConfig, HifiGanEngine
    pipe = AudioLDMPipeline(VAEDecoderEngine.random_init(VAEConfig.audioldm(), device, 1, dtype=DIFF_DT),
                            ClapTextEngine.random_init(ClapTextConfig(), device, 2, dtype=DIFF_DT),
                            None, UNetEngine.random_init(UNetConfig.audioldm_l(), device, 3, dtype=DIFF_DT), DDIMScheduler(beta_start=0.0015, beta_end=0.0195),
                            HifiGanEngine.random_init(HifiGanConfig.audioldm(), device, 4, dtype=DIFF_DT))
    ids = torch.randint(3, 50000, (1, 12)); ids[0, 0] = 0; ids[0, -1] = 2
    emb = pipe.text_encoder.text_embeds(torch.cat([ids, ids]), normalize=True)
    call = lambda: pipe(prompt_embeds=emb[1:], negative_prompt_embeds=emb[:1], audio_length_in_s=5.0, num_inference_steps=40,
                        guidance_scale=2.5, generator=torch.Generator(device=device).manual_seed(0))
    call()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    call()
    torch.cuda.synchronize(device)
    out["audio_clip_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    x = torch.randn(2, 125, 16, 8, generator=g, device=device).to(DIFF_DT)
    pipe.unet.step(x, 0)
    out["audio_unet_step_ms"] = round(_ev_ms(lambda i: pipe.unet.step(x, i), 10, device), 3)
    out["audio_clip"] = "AudioLDM-L shapes, 5.0 s, 40 DDIM steps, CFG, mel VAE + HiFi-GAN; CLAP text (2 prompts) outside"
    del pipe
    torch.cuda.empty_cache()
    return out


def llama8b_numbers(device):
    """BASELINE configs[0]/[2]: DeepSeek-R1-Distill-Llama-8B shapes, 256-token prompt + 256 greedy tokens."""
    from spider_amd.llm import LlamaEngine, LLMConfig
    cfg = LLMConfig.llama3_8b()
    eng = LlamaEngine.random_init(cfg, device, max_batch=1, max_len=600, seed=0)
    ids = torch.randint(3, cfg.vocab, (1, 256), generator=torch.Generator(device=device).manual_seed(1), device=device)
    eng.generate(input_ids=ids, max_new_tokens=8, sync_every=8)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    eng.generate(input_ids=ids, max_new_tokens=2, use_graph=False)
    torch.cuda.synchronize(device)
    tp = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.generate(input_ids=ids, max_new_tokens=256, sync_every=256)
    torch.cuda.synchronize(device)
    tg = time.perf_counter() - t0
    tok_s = 254 / max(tg - tp, 1e-6)
    wbytes = 2 * (cfg.layers * (cfg.hidden * (cfg.n_q + 2 * cfg.n_kv) * cfg.head_dim + cfg.n_q * cfg.head_dim * cfg.hidden + 3 * cfg.hidden * cfg.inter)
                  + cfg.vocab * cfg.hidden)
    kv = 2 * cfg.layers * cfg.n_kv * cfg.head_dim * 2 * (256 + 128)
    del eng
    torch.cuda.empty_cache()
    return {"prefill_256_ms": round(tp * 1e3, 1), "decode_tokens_per_s": round(tok_s, 1),
            "decode_hbm_frac": round((wbytes + kv) * tok_s / 1e9 / HBM_PEAK_GBS, 4), "weight_bytes_per_token": wbytes}


def run_timed(resp, args, rank, world, device):
    """The contract's timed region, shared by every workload (and driven on gloo with a stand-in responder by
    tests/test_dp_gloo.py): W untimed warm-up steps, then EXACTLY K steps between barrier + device synchronize on both sides;
    a step = this rank's responses + the ONE gather of their padded outputs to rank 0; time = MAX over ranks. Afterwards every
    rank leaves the process group, BEFORE rank 0's secondary timings and CPU baseline (tens of seconds), so that no rank waits in a
    collective on a slow host. Returns (seconds, rank 0's last gathered payload | None, {"world_size", "backend"})."""
    is_cuda = getattr(device, "type", "cpu") == "cuda"

    def sync():
        if is_cuda:
            torch.cuda.synchronize(device)

    def one_step():
        return dp_mod().gather_padded(resp.respond(), args.batch, rank, world, dst=0)

    g = None
    for _ in range(args.warmup):
        g = one_step()
    if world > 1:
        dist.barrier()
    sync()
    if hasattr(resp, "step_log"):
        resp.step_log.clear()
    walls = []                      # host wall time of every timed step (a step ends with both streams synchronised)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        g = one_step()
        walls.append((time.perf_counter() - ts) * 1e3)
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    resp.step_wall_ms = walls
    dist_info = {"world_size": 1, "backend": None}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend()}
        dist.barrier()
        dist.destroy_process_group()
    return dt, g, dist_info


def dp_mod():
    from spider_amd import dp
    return dp


def _dist4(xs):
    xs = sorted(float(x) for x in xs)
    if not xs:
        return None
    q = lambda f: xs[min(len(xs) - 1, int(round(f * (len(xs) - 1))))]
    return {"min": round(xs[0], 2), "median": round(q(0.5), 2), "p90": round(q(0.9), 2), "max": round(xs[-1], 2), "n": len(xs)}


def step_statistics(resp):
    """What the timed region looked like, step by step (rank 0): the distribution of the steps' wall times, and per step the device
    span of each pass on its own stream (HIP events recorded by SpiderFreeInfer), the idle time of each stream inside the step
    (= wall - span: the stream had nothing to run), and the host wall time of the pieces of the LLM pass. Medians over the K timed
    steps, so one odd step cannot describe the run."""
    walls, log = getattr(resp, "step_wall_ms", []), list(getattr(resp, "step_log", []))
    out = {"step_wall_ms": _dist4(walls)}
    if log and len(log) == len(walls) and len(walls) <= 32:      # every timed step: wall, the two device spans, host time of generate
        out["steps"] = [[round(w, 1), r.get("llm_pass_ms"), r.get("decoder_pass_ms"), r.get("llm_generate_host_ms"), r.get("device_allocs"),
                         r.get("device_frees")] for w, r in zip(walls, log)]
        out["steps_columns"] = ["wall_ms", "llm_pass_ms", "decoder_pass_ms", "llm_generate_host_ms", "device_allocs_so_far", "device_frees_so_far"]
    keys = sorted({k for r in log for k, v in r.items() if isinstance(v, (int, float)) and not isinstance(v, bool)})
    out["per_step_median"] = {k: _dist4([r[k] for r in log if k in r])["median"] for k in keys}
    if log and len(log) == len(walls):
        idle = {"stream_llm_idle_ms": [w - r["llm_pass_ms"] for w, r in zip(walls, log) if "llm_pass_ms" in r],
                "stream_decoder_idle_ms": [w - r["decoder_pass_ms"] for w, r in zip(walls, log) if "decoder_pass_ms" in r]}
        out["gpu_idle_ms"] = {k: _dist4(v) for k, v in idle.items() if v}
        # host work outside both device spans: a step's wall minus the longer of its two spans (chat inputs, batch_decode,
        # tokenizer, PIL, thread start / join, the gather)
        crit = [w - max(r.get("llm_pass_ms", 0.0), r.get("decoder_pass_ms", 0.0)) for w, r in zip(walls, log)
                if "llm_pass_ms" in r or "decoder_pass_ms" in r]
        if crit:
            out["wall_minus_longest_span_ms"] = _dist4(crit)
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # self-launch, decided before any GPU call
        raise SystemExit(launch_ranks(args))
    from spider_amd import dp
    rank, world, local = dp.init_from_env()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback on the product path)"
    if os.environ.get("SPIDER_SHARE_GPU"):      # rehearsal: every rank computes on cuda:0 (with SPIDER_DIST_BACKEND=gloo)
        local = 0
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    a2m = args.workload == "any2many"
    resp = AnyToManyResponder(args, device, rank) if a2m else Responder(args, device, rank)

    dt, g, dist_info = run_timed(resp, args, rank, world, device)

    if rank == 0:
        a = args
        total = world * args.batch * args.steps
        base = {"value": round(total / dt, 4), "unit": "responses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16", "diffusion_dtype": args.diffusion_dtype, "unet_residual_stream": "fp32" if not args.no_stream32 else "16-bit",
                "unet_precise": getattr(args, "precise", 0), "data": "synthetic", "dist": dist_info}
        if g is not None:
            base["gathered"] = {k: list(v.shape) for k, v in g.items()}
            base["gather_bytes_per_rank"] = int(sum(v[0].numel() * v[0].element_size() for v in g.values()))
        if a2m:
            line = {"metric": "multimodal responses/sec (text->text+image+audio+video)", **base,
                    "config": {"workload": f"any-to-many (BASELINE configs[4]) through spider_amd.SpiderFreeInfer: {a.llm} text-decoder shapes, prompt {a.prompt_len} + {a.new_tokens} "
                                           "greedy tokens, routing of IMAGE+AUDIO+VIDEO tags through SpiderDecoder.generate_batch: SD-v1.5 512^2 "
                                           "(41 UNet calls), AudioLDM-L 5 s (40 steps + mel VAE + HiFi-GAN), zeroscope 16x320x576 (40 steps + VAE); "
                                           "one gather of the padded outputs to rank 0 per step",
                               "product_class": "spider_amd.SpiderFreeInfer", "schedule": a.schedule,
                               "prompts_per_gpu": a.batch, "parallelism": f"dp{world}", "timed_region_includes": resp.includes(),
                               "weights": "random-init of the true shapes"},
                    "rank0_stage_ms_last_step": {**resp.stage, **(resp.overlap_ms or {})}, "roofline": None, "cpu_baseline": None}
            print(json.dumps(line), flush=True)
        else:
            roof = measure_roofline(resp, device)
            sched_depth = resp.infer.depth
            extra = text_image_extras(args, resp, device)
            if not args.no_extras and not args.headline_only and world == 1:
                del resp
                torch.cuda.empty_cache()
                extra.update(other_decoders(device))
                extra["llama3_8b"] = llama8b_numbers(device)
                if not args.no_any2many:
                    try:        # a secondary figure must never cost the headline line
                        extra["any2many"] = any2many_extra(args, device)
                    except Exception as e:      # noqa: BLE001
                        extra["any2many"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            cpu = None if (args.no_cpu_baseline or args.headline_only or world > 1) else cpu_baseline(args)   # reported at N=1 only
            extra["roofline_response"] = response_roofline(args, extra, base["ms_per_step"])
            line = {"metric": "multimodal responses/sec (text->text+image)", **base,
                    "config": {"workload": f"SpiderFree text->text+1x512^2 image through spider_amd.SpiderFreeInfer (qwen2.5omni_spider_web.py:458-521): "
                                           f"{args.llm} text-decoder shapes, prompt {a.prompt_len} + {a.new_tokens} greedy tokens, batch_decode, "
                                           f"extract_answer, SpiderDecoderInfer -> SpiderDecoder.generate -> StableDiffusionPipeline: CLIP text encoder, SD-v1.5 UNet "
                                           f"64x64 latent, PNDM {a.denoise_steps} steps ({a.denoise_steps + 1} UNet calls), CFG batch 2, guidance 7.5, VAE decode, PIL",
                               "product_class": "spider_amd.SpiderFreeInfer",
                               "prompt_sharding": "global list sorted by expected length (dp.order_by_length), then strided over the ranks",
                               "prompts_per_gpu": a.batch, "parallelism": f"dp{world}", "timed_region_includes": extra.pop("_includes"),
                               "schedule": ((f"overlap (SpiderFreeInfer.submit, depth {sched_depth}): every step = ONE prompt pass + ONE decode loop + ONE decoder pass of "
                                             "consecutive independent requests on two HIP streams -- "
                                             + ("stream L: decode loop of request k+1; stream U: decoder pass of request k, then the prompt pass of request "
                                                "k+2 into the other KV cache set; a step returns the request submitted two steps earlier"
                                                if sched_depth == 3 else
                                                "stream L: LLM pass of request k+1; stream U: decoder pass of request k; a step returns the request "
                                                "submitted one step earlier")
                                             + "; the passes the first timed steps consume were made during warm-up and the last steps' passes are read by "
                                             "nobody: balanced, K of each pass in K timed steps; serial_ms_per_response is the one-stream latency of a "
                                             "single request (SpiderFreeInfer.predict)")
                                            if a.schedule == "overlap" else "serial (SpiderFreeInfer.predict): the two passes of a request back to back on one stream"),
                               "weights": "random-init of the true shapes"},
                    "roofline": roof, "cpu_baseline": cpu, **extra}
            print(json.dumps(line), flush=True)


def any2many_extra(args, device):
    """BASELINE configs[3]/[4] on this GPU, outside the timed region: ONE timed step (after one warm-up step) of the any-to-many
    workload at 8 prompts per GPU -- text + 512^2 image + 5 s audio + 16-frame video per response through the same product class --
    so that the driver's N=1 line carries a number for it. Stage times are device-event times of the pipeline calls (SpiderDecoder.
    stage_ms_device) and of the two passes, not host walls."""
    import copy
    a = copy.copy(args)
    a.workload, a.batch, a.throughput_batch = "any2many", 8, 0
    resp = AnyToManyResponder(a, device, 0)
    resp.respond()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out = resp.respond()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    res = {"responses_per_s": round(a.batch / dt, 4), "ms_per_step": round(dt * 1e3, 1), "prompts_per_gpu": a.batch, "steps_timed": 1,
           "schedule": a.schedule, "outputs": {k: list(v.shape) for k, v in out.items()},
           "rank0_stage_ms": {**resp.stage, **(resp.overlap_ms or {})},
           "workload": "text -> text + 512^2 image (SD-v1.5, 41 UNet calls) + 5 s audio (AudioLDM-L, 40 steps, mel VAE, HiFi-GAN) + 16x320x576 video "
                       "(zeroscope, 40 steps, VAE) per response; spider_amd.SpiderFreeInfer -> SpiderDecoder.generate_batch"}
    del resp
    torch.cuda.empty_cache()
    return res


def response_roofline(args, extra, ms_per_step):
    """Composite roofline of ONE response: the sum of every stage's algorithmic floor -- decode bytes / HBM peak,
    prefill / UNet / VAE flops / MFMA peak -- over the measured time of the step. The per-kernel objects (`roofline`,
    `roofline_unet_*`, `roofline_prefill_gemm`) say how close the best kernels are; this one says how close the RESPONSE is."""
    from spider_amd.llm import LLMConfig
    c = getattr(LLMConfig, args.llm)()
    params = c.layers * (c.hidden * (c.n_q + 2 * c.n_kv) * c.head_dim + c.n_q * c.head_dim * c.hidden + 3 * c.hidden * c.inter) + c.vocab * c.hidden
    S, T, B = args.prompt_len, args.new_tokens, args.batch
    kv_per_tok = 2 * c.layers * c.n_kv * c.head_dim * 2
    decode_bytes = T * 2 * params + B * sum(kv_per_tok * (S + t) for t in range(T))            # weights once per step, KV per row
    prefill_flops = B * (2 * params * S + 4 * c.layers * c.n_q * c.head_dim * S * S / 2)
    unet_flops_total = B * (args.denoise_steps + 1) * 2 * extra["unet_flops_per_sample"]["total"] * 1e9
    vae_flops = B * 1.27e12                                                                    # AutoencoderKL decode at 512^2
    floors = {"decode_ms": decode_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, "prefill_ms": prefill_flops / (MFMA_PEAK_TF * 1e12) * 1e3,
              "unet_ms": unet_flops_total / (MFMA_PEAK_TF * 1e12) * 1e3, "vae_ms": vae_flops / (MFMA_PEAK_TF * 1e12) * 1e3}
    floor = sum(floors.values())
    # with the LLM pass and the decoder pass on different streams the two halves could hide each other completely: the floor of a
    # step of the overlapped schedule is the larger half, not the sum (both are reported; `frac` uses the conservative sum)
    overlap_floor = max(floors["decode_ms"] + floors["prefill_ms"], floors["unet_ms"] + floors["vae_ms"])
    return {"floor_ms": round(floor, 1), "measured_ms": ms_per_step, "frac": round(floor / ms_per_step, 4),
            "overlap_floor_ms": round(overlap_floor, 1), "serial_measured_ms": extra.get("serial_ms_per_response"),
            "serial_frac": round(floor / extra["serial_ms_per_response"], 4) if extra.get("serial_ms_per_response") else None,
            "floors_ms": {k: round(v, 1) for k, v in floors.items()},
            "algorithmic": {"decode_hbm_bytes": int(decode_bytes), "prefill_flops": prefill_flops, "unet_flops": unet_flops_total, "vae_flops": vae_flops},
            "peaks": {"hbm_GBps": HBM_PEAK_GBS, "mfma_TFLOPs": MFMA_PEAK_TF}}


def text_image_extras(args, resp, device):
    """Secondary timings of the headline workload on rank 0 (outside the timed region)."""
    from spider_amd import ops
    from spider_amd.unet import unet_flops, UNetConfig
    a = args
    extra = {"_includes": resp.includes()}
    extra["timed_steps"] = step_statistics(resp)              # per-step distribution + per-stream spans / idle time (medians)
    if resp.overlap_ms:
        extra["overlap_last_step"] = resp.overlap_ms      # device time of the two concurrent passes of the last timed step
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # UNet step ms (BASELINE.json's second metric): graph replay of one CFG-batch-2 evaluation
    x2 = _unet_input(resp.unet, resp.latents0[:1].contiguous())
    ts = resp.sched.set_timesteps(a.denoise_steps)
    resp.unet.prepare(ts, resp.enc_synth[:2].contiguous())
    resp.unet.step(x2, 0)
    torch.cuda.synchronize(device)
    e0.record()
    for i in range(10):
        resp.unet.step(x2, i % len(ts))
    e1.record(); e1.synchronize()
    unet_ms = e0.elapsed_time(e1) / 10
    fl = unet_flops(UNetConfig.sd15(), 64, 64)
    extra["unet_step_ms"] = round(unet_ms, 3)
    extra["unet_tflops_per_s"] = round(2 * fl["total"] / (unet_ms * 1e-3) / 1e12, 1)
    extra["unet_flops_per_sample"] = {k: round(v / 1e9, 2) for k, v in fl.items()}
    c = resp.llm.cfg
    wbytes = 2 * (c.layers * (c.hidden * (c.n_q + 2 * c.n_kv) * c.head_dim + c.n_q * c.head_dim * c.hidden + 3 * c.hidden * c.inter) + c.vocab * c.hidden)
    kvb = 2 * c.layers * c.n_kv * c.head_dim * 2 * (a.prompt_len + a.new_tokens // 2)

    def llm_phase(B):
        ids = resp.prompt[:B].contiguous()
        resp.llm.generate(input_ids=ids, max_new_tokens=4, sync_every=4)          # graph for this batch captured outside the timing
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        resp.llm.generate(input_ids=ids, max_new_tokens=2, use_graph=False)
        torch.cuda.synchronize(device)
        t_prefill = time.perf_counter() - t1
        t1 = time.perf_counter()
        resp.llm.generate(input_ids=ids, max_new_tokens=a.new_tokens, sync_every=a.new_tokens)
        torch.cuda.synchronize(device)
        t_gen = time.perf_counter() - t1
        tok_s = B * (a.new_tokens - 2) / max(t_gen - t_prefill, 1e-6)
        # bytes that must cross HBM per decoded token-step: the weights once per step (shared by the B rows) + every row's KV
        frac = (wbytes + B * kvb) * (tok_s / B) / 1e9 / HBM_PEAK_GBS
        return t_prefill, tok_s, frac

    if a.headline_only:
        extra["unet_step_mfma_frac"] = round(extra["unet_tflops_per_s"] / MFMA_PEAK_TF, 4)
        return extra
    # one response start to finish on ONE stream (the latency of a single request; `value` counts a step of the overlapped
    # schedule when --schedule overlap)
    resp.respond_serial()
    torch.cuda.synchronize(device)
    resp.step_log.clear()
    sw = []
    for _ in range(5):
        t1 = time.perf_counter()
        resp.respond_serial()
        torch.cuda.synchronize(device)
        sw.append((time.perf_counter() - t1) * 1e3)
    extra["serial_ms_per_response"] = round(sorted(sw)[len(sw) // 2], 2)          # median of 5
    extra["serial_responses_per_s"] = round(a.batch / (extra["serial_ms_per_response"] * 1e-3), 4)
    resp.step_wall_ms = sw
    ser = step_statistics(resp)
    # the same host / GPU split for ONE request on one stream: device spans of its two passes (events), stage times inside the
    # decoder pass (device), and what is left of the wall time = host work on the critical path + launch gaps
    extra["serial_split"] = {"wall_ms": ser["step_wall_ms"], "per_response_median": ser["per_step_median"],
                             "decoder_stage_ms_device_last": resp.stage,
                             "wall_minus_device_spans_ms": _dist4([w - r.get("llm_pass_ms", 0.0) - r.get("decoder_pass_ms", 0.0)
                                                                   for w, r in zip(sw, resp.step_log)])}
    if a.schedule == "overlap" and not getattr(a, "precise", 0):
        try:            # a secondary figure must never cost the headline line
            extra["precise_mode"] = precise_mode_leg(a, resp, device)
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc()
            extra["precise_mode"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    tp, tok_s, frac = llm_phase(a.batch)
    extra["llm_prefill_ms"] = round(tp * 1e3, 1)
    extra["llm_decode_tokens_per_s"] = round(tok_s, 1)
    extra["llm_decode_hbm_frac"] = round(frac, 4)
    if a.throughput_batch and a.throughput_batch != a.batch:
        tb = min(a.throughput_batch, 8)
        _, tok8, frac8 = llm_phase(tb)
        extra["llm_decode_batched"] = {"rows": tb, "tokens_per_s": round(tok8, 1), "hbm_frac": round(frac8, 4)}
        resp.respond(tb)                           # serial + pipeline primed (graph captures)
        resp.respond(tb)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(2):
            resp.respond(tb)
        torch.cuda.synchronize(device)
        dtb = (time.perf_counter() - t1) / 2
        curve = []
        for b2 in (2, 4):                          # the same measurement at 2 and 4 prompts per GPU (requests answered together)
            if b2 >= tb:
                continue
            resp.respond(b2)
            resp.respond(b2)
            torch.cuda.synchronize(device)
            t2 = time.perf_counter()
            for _ in range(2):
                resp.respond(b2)
            torch.cuda.synchronize(device)
            d2 = (time.perf_counter() - t2) / 2
            curve.append({"prompts_per_gpu": b2, "responses_per_s": round(b2 / d2, 4), "ms_per_batch": round(d2 * 1e3, 1)})
        extra["batched_throughput_curve"] = curve + [{"prompts_per_gpu": tb, "responses_per_s": round(tb / dtb, 4), "ms_per_batch": round(dtb * 1e3, 1)}]
        extra["batched_throughput"] = {"prompts_per_gpu": tb, "responses_per_s": round(tb / dtb, 4), "ms_per_batch": round(dtb * 1e3, 1),
                                       "schedule": a.schedule,
                                       "note": "same workload, independent prompts batched on one GPU (BASELINE config 5 uses 8 per GPU), same "
                                               "schedule as the headline (a step = one batched LLM pass + one batched decoder pass); not the "
                                               "headline value"}
    extra["roofline_prefill_gemm"] = measure_prefill_gemm_roofline(device, resp.llm.cfg, a.prompt_len)
    extra["roofline_unet_conv"] = measure_mfma_roofline(device)
    extra["roofline_unet_attention"] = measure_attention_roofline(device)
    extra["unet_step_mfma_frac"] = round(extra["unet_tflops_per_s"] / 2500.0, 4)
    extra["mfma_peak_note"] = ("every MFMA 'frac' is against the nominal 2.5 PFLOP/s (2.4 GHz); MI355X_MICROARCH.md (DVFS give-back, item 1) records "
                               "1.90-1.95 GHz under a saturated bf16 GEMM on random operands (1,247 TFLOP/s vs 1,483 on zeros), i.e. ~0.50 of "
                               "nominal is the real-data ceiling of a matrix-core-bound loop on this part")
    return extra


if __name__ == "__main__":
    main()
