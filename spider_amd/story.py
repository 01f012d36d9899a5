"""StoryDiffusion on the HIP engines: SDXL pipeline + Consistent Self-Attention + story_generation, as Spider
drives it (StoryDiffusion/Comic_Generation.py; callers spider_decoder_infer.py:71-83, demo/inference_api.py:144).

Reference behaviour kept:
  * processor schedule SpatialAttnProcessor2_0.__call__ (Comic_Generation.py:74-127): write-mode id-bank per step,
    plain attention for cur_step < 5, then Bernoulli(0.7 / 0.9 from step 20) consistent attention, mask slices
    [:4N,:4N] (write) / rows [4N:] (read), step counter advanced when all `total_count` processors have run, masks
    regenerated every step (cal_attn_mask_xl, utils/gradio_utils.py:241-287)
  * story_generation (Comic_Generation.py:320-467): only `up_blocks.*.attn1` get the processor; 768x768, 50 DDIM
    steps, guidance 5.0, id_length 4, sa32 = sa64 = 0.5, seed 2047, FreeU(0.6, 0.4, 1.1, 1.2) (:315); prompts =
    general + "," + prompt; style templates; `styles.get(name, "(No style)")` fallback; returns id + real images
MI355X-first differences:
  * module globals (:82-84) -> explicit StoryState; coin flips / uniforms are injectable for parity tests
  * the [4N,4N] bool mask is never built: the mask is column-structured, so the kernel takes a 64-bit-packed keep
    vector + the image block length (spider_attn_bf16 keep_bits / blk / q_off)
  * consistent attention = the [8,N,C] -> [2,4N,C] reshape is a free view; q,k,v come from one fused GEMM
  * PNGs are not written to an absolute path (side effect of :414-422,442,449 is optional via save_dir)
"""
from __future__ import annotations

import json
import os
import random
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import ops
from .clip import CLIPTextEngine
from .pipelines import PipelineOutput, numpy_to_pil
from .schedulers import DDIMScheduler
from .unet import UNetEngine, denoise
from .vae import VAEDecoderEngine

BF16 = torch.bfloat16

# Style templates are prompt DATA of upstream StoryDiffusion (utils/style_template.py: name -> (positive template with
# "{prompt}", negative prompt)). They are not shipped here; point SPIDER_STORY_STYLES at a JSON file
# {"name": ["positive {prompt} ...", "negative ..."], ...} or call register_styles(). "(No style)" -- the entry the
# reference falls back to for unknown names (Comic_Generation.py:408-413) -- is built in.
_STYLES: Dict[str, tuple] = {"(No style)": ("{prompt}", "")}
_STYLES_LOADED = False


def register_styles(table: Dict[str, tuple]) -> None:
    _STYLES.update({k: tuple(v) for k, v in table.items()})


def styles() -> Dict[str, tuple]:
    global _STYLES_LOADED
    if not _STYLES_LOADED:
        _STYLES_LOADED = True
        path = os.environ.get("SPIDER_STORY_STYLES")
        if path and os.path.exists(path):
            with open(path) as f:
                register_styles(json.load(f))
    return _STYLES


def setup_seed(seed: int):
    """Comic_Generation.py:35-40"""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def pack_keep_bits(keep: torch.Tensor) -> torch.Tensor:
    """bool [L] (host) -> int64 words, bit j%64 of word j//64 = keep[j] (layout of spider_attn_bf16's keep_bits)."""
    k = keep.to(torch.bool).cpu().numpy()
    L = (len(k) + 63) // 64 * 64
    pad = np.zeros(L, dtype=np.uint8)
    pad[: len(k)] = k
    words = np.packbits(pad.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)
    return torch.from_numpy(words.view(np.int64).copy())


@dataclass
class StoryState:
    total_count: int
    height: int
    width: int
    id_length: int = 4
    sa32: float = 0.5
    sa64: float = 0.5
    write: bool = True
    cur_step: int = 0
    attn_count: int = 0
    coin: Callable[[], float] = random.random
    uniforms: Optional[Callable[[int], torch.Tensor]] = None    # n -> [n] uniforms in [0,1) (host or device)
    keep1024: Optional[torch.Tensor] = None                      # column keep vectors (bool, host)
    keep4096: Optional[torch.Tensor] = None
    bits: Dict[tuple, torch.Tensor] = field(default_factory=dict)

    @property
    def total_length(self):
        return self.id_length + 1

    def regen_masks(self, device):
        """cal_attn_mask_xl reduced to its information content: the keep vector of each resolution (the per-row
        'own block' override is applied inside the kernel)."""
        n1, n4 = (self.height // 32) * (self.width // 32), (self.height // 16) * (self.width // 16)
        gen = self.uniforms or (lambda n: torch.rand(n, device=device))
        k1 = (gen(self.total_length * n1).reshape(-1).cpu() < self.sa32)
        k4 = (gen(self.total_length * n4).reshape(-1).cpu() < self.sa64)
        k1[self.id_length * n1:] = False
        k4[self.id_length * n4:] = False
        self.keep1024, self.keep4096 = k1, k4
        self.bits = {}

    def keep_bits(self, use1024: bool, n_keys: int, device) -> torch.Tensor:
        key = (use1024, n_keys)
        if key not in self.bits:
            k = self.keep1024 if use1024 else self.keep4096
            self.bits[key] = pack_keep_bits(k[:n_keys]).to(device)
        return self.bits[key]


class ConsistentSelfAttention:
    """UNetEngine.self_attn_hook: replaces attn1 of every up-block transformer (Comic_Generation.py:353-371)."""

    def __init__(self, state: StoryState):
        self.st = state
        self.id_bank: Dict[str, Dict[int, List[torch.Tensor]]] = {}

    @staticmethod
    def wants(name: str) -> bool:
        return name.startswith("up_blocks") and name.endswith("attn1")

    @staticmethod
    def count_processors(unet: UNetEngine) -> int:
        return sum(1 for k in unet.w if k.startswith("up_blocks") and k.endswith(".attn1.qkv"))

    def __call__(self, eng: UNetEngine, name: str, y: torch.Tensor, heads: int) -> torch.Tensor:
        st = self.st
        L = st.id_length
        B, N, C = y.shape
        b = name[: -len(".attn1")]
        wqkv = eng.w[b + ".attn1.qkv"]
        bank = self.id_bank.setdefault(name, {})
        enc = None
        if st.write:
            bank[st.cur_step] = [y[:L].clone(), y[L:].clone()]
        else:
            enc = torch.cat((bank[st.cur_step][0], y[:1], bank[st.cur_step][1], y[1:])).view(2, (L + 1) * N, C)

        def plain(enc_):
            if enc_ is None:
                qkv = ops.gemm(y, wqkv)
                return ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
            q = ops.gemm(y, wqkv[:C])
            kv = ops.gemm(enc_, wqkv[C:])
            return ops.attention(q, kv[..., :C], kv[..., C:], heads)

        if st.cur_step < 5:
            out = plain(enc)
        else:
            r = st.coin()
            thr = 0.3 if st.cur_step < 20 else 0.1
            if r > thr:
                use1024 = N == (st.height // 32) * (st.width // 32)
                if st.write:
                    img = B // 2
                    x = y.view(2, img * N, C)                       # [8,N,C] -> [2,4N,C]: a view, no copy
                    qkv = ops.gemm(x, wqkv)
                    o = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads,
                                      keep_bits=st.keep_bits(use1024, img * N, y.device), blk=N, q_off=0)
                    out = o.view(B, N, C)
                else:
                    q = ops.gemm(y, wqkv[:C])
                    kv = ops.gemm(enc, wqkv[C:])
                    out = ops.attention(q, kv[..., :C], kv[..., C:], heads,
                                        keep_bits=st.keep_bits(use1024, (L + 1) * N, y.device), blk=N, q_off=L * N)
            else:
                out = plain(None)
        st.attn_count += 1
        if st.attn_count == st.total_count:
            st.attn_count = 0
            st.cur_step += 1
            st.regen_masks(y.device)
        return out


class StableDiffusionXLPipeline:
    """SDXL text-to-image on the HIP engines (two text encoders, text_time added conditioning, DDIM, FreeU)."""
    vae_scale_factor = 8

    def __init__(self, unet: UNetEngine, vae: VAEDecoderEngine, text_encoder: CLIPTextEngine, text_encoder_2: CLIPTextEngine,
                 tokenizer, tokenizer_2, scheduler=None, default_size=1024):
        self.unet, self.vae = unet, vae
        self.text_encoder, self.text_encoder_2, self.tokenizer, self.tokenizer_2 = text_encoder, text_encoder_2, tokenizer, tokenizer_2
        self.scheduler = scheduler or DDIMScheduler()
        self.default_size = default_size
        self.device = unet.device
        self.vae_scale_factor = 2 ** (len(vae.cfg.block_out) - 1)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device="cuda:0", **unused):
        from transformers import CLIPTokenizer
        from .vae import VAEConfig
        vae = VAEDecoderEngine.from_pretrained(os.path.join(path, "vae"), device)
        vcfg = json.load(open(os.path.join(path, "vae", "config.json")))
        vae.cfg.scaling = vcfg.get("scaling_factor", 0.13025)
        vae.w["post_quant_conv.weight_scaled"] = (vae.w["post_quant_conv.weight"].float() / vae.cfg.scaling).to(BF16).contiguous()
        return cls(UNetEngine.from_pretrained(os.path.join(path, "unet"), device), vae,
                   CLIPTextEngine.from_pretrained(os.path.join(path, "text_encoder"), device),
                   CLIPTextEngine.from_pretrained(os.path.join(path, "text_encoder_2"), device),
                   CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer")),
                   CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer_2")))

    def to(self, *a, **k):
        return self

    def enable_freeu(self, s1, s2, b1, b2):
        self.unet.freeu = (s1, s2, b1, b2)

    def _encode(self, prompts: List[str]):
        embs, pooled = [], None
        for tk, te in ((self.tokenizer, self.text_encoder), (self.tokenizer_2, self.text_encoder_2)):
            ids = tk(prompts, padding="max_length", max_length=tk.model_max_length, truncation=True, return_tensors="pt").input_ids
            o = te.encode(ids, return_all=True)
            embs.append(o["penultimate"])
            pooled = o["pooled"]                       # the second encoder's projected pooled state is kept
        return torch.cat(embs, -1).contiguous(), pooled

    @torch.no_grad()
    def __call__(self, prompt, num_inference_steps=50, guidance_scale=5.0, height=None, width=None, negative_prompt=None,
                 generator=None, latents=None, output_type="pil", **unused):
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        B = len(prompt)
        height, width = height or self.default_size, width or self.default_size
        neg = [negative_prompt or ""] * B if not isinstance(negative_prompt, (list, tuple)) else list(negative_prompt)
        pe, pp = self._encode(prompt)
        ne, npool = self._encode(neg)
        enc = torch.cat([ne, pe]).contiguous()                                      # [uncond | cond]
        tid = torch.tensor([[height, width, 0, 0, height, width]] * (2 * B), dtype=torch.float32)
        added = dict(text_embeds=torch.cat([npool, pp]).float(), time_ids=tid)
        shape = (B, self.unet.cfg.in_ch, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else self.device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(self.device)
        lat = denoise(self.unet, self.scheduler, latents.to(self.device, torch.float32), enc, guidance_scale,
                      num_inference_steps, added=added)
        if output_type == "latent":
            return PipelineOutput(lat)
        img = self.vae.decode(lat).cpu().permute(0, 2, 3, 1).float().numpy()
        return PipelineOutput(numpy_to_pil(img) if output_type == "pil" else img)


def init_story_generation(model_path: str, device="cuda:0") -> StableDiffusionXLPipeline:
    """Comic_Generation.py:297-318: load SDXL, FreeU on, DDIM with 50 steps."""
    pipe = StableDiffusionXLPipeline.from_pretrained(model_path, device=device)
    pipe.enable_freeu(s1=0.6, s2=0.4, b1=1.1, b2=1.2)
    pipe.scheduler = DDIMScheduler()
    pipe.scheduler.set_timesteps(50)
    return pipe


NEGATIVE_PROMPT = ("naked, deformed, bad anatomy, disfigured, poorly drawn face, mutation, extra limb, ugly, disgusting, "
                   "poorly drawn hands, missing limb, floating limbs, disconnected limbs, blurry, watermarks, oversaturated, "
                   "distorted hands, amputation")


def story_generation(pipe, general_prompt=None, prompt_array=None, style_name=None, height=768, width=768, num_steps=50,
                     guidance_scale=5.0, seed=2047, id_length=4, sa32=0.5, sa64=0.5, save_dir: Optional[str] = None,
                     state_hooks: Optional[dict] = None, output_type="pil"):
    """Comic_Generation.py:320-467. Returns id_images + real_images."""
    DEFAULT_STYLE_NAME = "(No style)"
    unet = pipe.unet
    st = StoryState(total_count=ConsistentSelfAttention.count_processors(unet), height=height, width=width,
                    id_length=id_length, sa32=sa32, sa64=sa64, write=False, **(state_hooks or {}))
    hook = ConsistentSelfAttention(st)
    unet.self_attn_hook = hook
    print("successsfully load consistent self-attention")
    print(f"number of the processor : {st.total_count}")
    try:
        st.regen_masks(unet.device)
        general_prompt = "a man with a black suit" if general_prompt is None else general_prompt
        if prompt_array is None:
            prompt_array = ["wake up in the bed", "have breakfast", "is on the road, go to the company", "work in the company",
                            "running in the playground", "reading book in the home"]
        sty = styles()
        def apply_style_positive(name, positive):
            p, n = sty.get(name, sty[DEFAULT_STYLE_NAME])
            return p.replace("{prompt}", positive)
        def apply_style(name, positives, negative=""):
            p, n = sty.get(name, sty[DEFAULT_STYLE_NAME])
            return [p.replace("{prompt}", positive) for positive in positives], n + " " + negative
        style_name = "Comic book" if style_name is None else style_name
        setup_seed(seed)
        generator = torch.Generator(device=unet.device).manual_seed(seed)
        prompts = [general_prompt + "," + p for p in prompt_array]
        id_prompts, real_prompts = prompts[:id_length], prompts[id_length:]
        st.write, st.cur_step, st.attn_count = True, 0, 0
        id_prompts, negative_prompt = apply_style(style_name, id_prompts, NEGATIVE_PROMPT)
        id_images = pipe(id_prompts, num_inference_steps=num_steps, guidance_scale=guidance_scale, height=height, width=width,
                         negative_prompt=negative_prompt, generator=generator, output_type=output_type).images
        st.write = False
        real_images = []
        for real_prompt in real_prompts:
            st.cur_step = 0
            rp = apply_style_positive(style_name, real_prompt)
            real_images.append(pipe(rp, num_inference_steps=num_steps, guidance_scale=guidance_scale, height=height, width=width,
                                    negative_prompt=negative_prompt, generator=generator, output_type=output_type).images[0])
        out = list(id_images) + real_images
        if save_dir is not None and output_type == "pil":
            os.makedirs(save_dir, exist_ok=True)
            for i, im in enumerate(out):
                im.save(os.path.join(save_dir, f"image_{i}.png"))
        return out
    finally:
        unet.self_attn_hook = None
