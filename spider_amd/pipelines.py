"""Diffusion pipeline objects with the reference's call contract (seam B4):

    pipe = StableDiffusionPipeline.from_pretrained(path, torch_dtype=...).to(device)
    pipe(prompt=[...] | prompt_embeds=T[B,77,C], guidance_scale, num_inference_steps, height, width,
         negative_prompt, generator, latents, output_type, return_prompts_only=False) -> .images
    pipe(captions, return_prompts_only=True) -> text-encoder states WITHOUT the CFG concat

following spider/models/custom_sd.py: __call__ :476-667, _encode_prompt :223-374 (left truncation :267-276),
prepare_latents :459-474, denoising loop :627-652, decode_latents :386-393. Everything numeric runs on the HIP
engines (CLIPTextEngine, UNetEngine, VAEDecoderEngine); this file is host glue. Differences from the reference,
all deliberate: models are loaded once per pipeline object (the reference reloads from disk in every decode call,
spider_decoder.py:109,114); latents stay fp32 in HBM; the safety checker (an output filter, not on the numeric
path) is not run.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Union

import numpy as np
import torch

from . import ops
from .clip import CLIPTextEngine
from .registry import registry
from .schedulers import SCHEDULERS, PNDMScheduler
from .unet import UNetEngine, denoise
from .vae import VAEDecoderEngine


class PipelineOutput:
    def __init__(self, images, nsfw_content_detected=None):
        self.images = images
        self.nsfw_content_detected = nsfw_content_detected


def numpy_to_pil(images: np.ndarray):
    from PIL import Image
    if images.ndim == 3:
        images = images[None]
    images = (images * 255).round().astype("uint8")
    return [Image.fromarray(im) for im in images]


@registry.register_model("sd")
class StableDiffusionPipeline:
    vae_scale_factor = 8

    def __init__(self, unet: UNetEngine, vae: Optional[VAEDecoderEngine], text_encoder: Optional[CLIPTextEngine], tokenizer,
                 scheduler=None, sample_size: int = 64):
        self.unet, self.vae, self.text_encoder, self.tokenizer = unet, vae, text_encoder, tokenizer
        self.scheduler = scheduler or PNDMScheduler()
        self.sample_size = sample_size
        self.device = unet.device
        if vae is not None:
            self.vae_scale_factor = 2 ** (len(vae.cfg.block_out) - 1)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device="cuda:0", **unused):
        """diffusers directory layout: unet/, vae/, text_encoder/, tokenizer/, scheduler/scheduler_config.json."""
        from transformers import CLIPTokenizer
        sc = json.load(open(os.path.join(path, "scheduler", "scheduler_config.json")))
        sched_cls = SCHEDULERS.get(sc.get("_class_name", "PNDMScheduler"), PNDMScheduler)
        sched = sched_cls(**{k: v for k, v in sc.items() if k in ("num_train_timesteps", "beta_start", "beta_end", "steps_offset")})
        ucfg = json.load(open(os.path.join(path, "unet", "config.json")))
        return cls(UNetEngine.from_pretrained(os.path.join(path, "unet"), device),
                   VAEDecoderEngine.from_pretrained(os.path.join(path, "vae"), device),
                   CLIPTextEngine.from_pretrained(os.path.join(path, "text_encoder"), device),
                   CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer")), sched, ucfg.get("sample_size", 64))

    def to(self, device=None, *a, **k):
        return self

    # ------------------------------------------------------------------ prompt encoding (custom_sd.py:223-374)
    def _tokenize(self, prompt: List[str]) -> torch.Tensor:
        tk = self.tokenizer
        untruncated = tk(prompt, padding="longest", return_tensors="pt").input_ids
        if untruncated.shape[-1] > tk.model_max_length:   # truncate from the LEFT (custom_sd.py:267-276)
            prompt = tk.batch_decode(untruncated[:, -1 - tk.model_max_length: -1])
        return tk(prompt, padding="max_length", max_length=tk.model_max_length, truncation=True, return_tensors="pt").input_ids

    def _encode_prompt(self, prompt, num_images_per_prompt, do_cfg, negative_prompt=None, prompt_embeds=None,
                       negative_prompt_embeds=None) -> torch.Tensor:
        if prompt is not None and isinstance(prompt, str):
            prompt = [prompt]
        batch_size = len(prompt) if prompt is not None else prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_embeds = self.text_encoder.encode(self._tokenize(prompt))
        prompt_embeds = prompt_embeds.to(device=self.device, dtype=torch.bfloat16)
        bs, seq, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, seq, -1)
        if do_cfg and negative_prompt_embeds is None:
            if negative_prompt is None:
                uncond = [""] * batch_size
            elif isinstance(negative_prompt, str):
                uncond = [negative_prompt]
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt` has batch size {len(negative_prompt)}, but `prompt` has batch size {batch_size}.")
            else:
                uncond = list(negative_prompt)
            ids = self.tokenizer(uncond, padding="max_length", max_length=prompt_embeds.shape[1], truncation=True,
                                 return_tensors="pt").input_ids
            negative_prompt_embeds = self.text_encoder.encode(ids)
        if do_cfg:
            n = negative_prompt_embeds.to(device=self.device, dtype=torch.bfloat16)
            n = n.repeat(1, num_images_per_prompt, 1).view(batch_size * num_images_per_prompt, n.shape[1], -1)
            prompt_embeds = torch.cat([n, prompt_embeds])   # [uncond | cond], one UNet batch (custom_sd.py:372)
        return prompt_embeds.contiguous()

    def prepare_latents(self, batch_size, channels, height, width, generator, latents=None):
        shape = (batch_size, channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else self.device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(self.device)
        else:
            latents = latents.to(self.device, torch.float32)
        return (latents * self.scheduler.init_noise_sigma).contiguous()

    def decode_latents(self, latents) -> np.ndarray:
        img = self.vae.decode(latents)                       # [B,3,H,W] fp32 in [0,1]
        return img.cpu().permute(0, 2, 3, 1).float().numpy()

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None,
                 num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True, callback=None,
                 callback_steps: int = 1, cross_attention_kwargs=None, return_prompts_only: bool = False):
        height = height or self.sample_size * self.vae_scale_factor
        width = width or self.sample_size * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0 and not return_prompts_only
        embeds = self._encode_prompt(prompt, num_images_per_prompt, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds)
        if return_prompts_only:
            return embeds
        lat = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.cfg.in_ch, height, width, generator, latents)
        lat = denoise(self.unet, self.scheduler, lat, embeds, guidance_scale, num_inference_steps)
        if output_type == "latent":
            return PipelineOutput(lat)
        image = self.decode_latents(lat)
        if output_type == "pil":
            image = numpy_to_pil(image)
        return PipelineOutput(image) if return_dict else (image, None)
