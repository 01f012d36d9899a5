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
from .clap import ClapTextEngine
from .clip import CLIPTextEngine
from .registry import registry
from .schedulers import PNDMScheduler, scheduler_from_config
from .unet import UNetEngine, denoise
from .unet3d import UNet3DEngine, video_denoise
from .vae import VAEDecoderEngine
from .vocoder import HifiGanEngine


def engine_dtype(torch_dtype):
    """The 16-bit engine format for a `from_pretrained(..., torch_dtype=...)` request. The reference always passes
    torch.float16 (spider_decoder.py:109,114,130,136,153,159; base_model.py:211); bfloat16 is honoured; None / float32
    (diffusers: keep the checkpoint precision) maps to float16, the closest format the MFMA engines have -- with a warning for
    an explicit float32 request: the f16 engines saturate above 65504 where fp32 (and bf16) would not; pass torch.bfloat16 for
    range instead of precision."""
    if torch_dtype == torch.float32:
        import warnings
        warnings.warn("spider_amd diffusion engines compute on 16-bit MFMA operands: torch_dtype=float32 runs as float16 (+ fp32 "
                      "residual stream); pass torch.bfloat16 if the checkpoint needs fp32's range", stacklevel=3)
    return torch.bfloat16 if torch_dtype == torch.bfloat16 else torch.float16


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
    def from_pretrained(cls, path: str, torch_dtype=None, device="cuda:0", precise: int = 0, **unused):
        """diffusers directory layout: unet/, vae/, text_encoder/, tokenizer/, scheduler/scheduler_config.json."""
        from transformers import CLIPTokenizer
        sc = json.load(open(os.path.join(path, "scheduler", "scheduler_config.json")))
        sched = scheduler_from_config(sc)
        ucfg = json.load(open(os.path.join(path, "unet", "config.json")))
        dt = engine_dtype(torch_dtype)
        # stream32: fp32 master of the residual stream (+ 2.9 % per UNet evaluation, 26 % closer to the fp32 oracle: DESIGN.md section 4);
        # precise = 1 / 2 (an extra keyword of this loader, 0 = off): the stream's consumers read the master -- inside north_star's 1e-3
        return cls(UNetEngine.from_pretrained(os.path.join(path, "unet"), device, dtype=dt, stream32=True, precise=precise),
                   VAEDecoderEngine.from_pretrained(os.path.join(path, "vae"), device, scaling=0.18215, dtype=dt),   # custom_sd.py:388
                   CLIPTextEngine.from_pretrained(os.path.join(path, "text_encoder"), device, dtype=dt),
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
        prompt_embeds = prompt_embeds.to(device=self.device, dtype=self.unet.dtype)
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
            n = negative_prompt_embeds.to(device=self.device, dtype=self.unet.dtype)
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


class AudioPipelineOutput:
    def __init__(self, audios):
        self.audios = audios


@registry.register_model("ad")
class AudioLDMPipeline:
    """AudioLDM text-to-audio with the reference's call contract (spider/models/custom_ad.py):

        pipe = AudioLDMPipeline.from_pretrained(path, torch_dtype=...).to(device)
        pipe(prompt=[...] | prompt_embeds=T[B,512], audio_length_in_s, num_inference_steps, guidance_scale,
             negative_prompt, num_waveforms_per_prompt, generator, latents, output_type="np",
             return_prompts_only=False) -> .audios   (np.ndarray [B, samples] at vocoder.sampling_rate)

    __call__ :400-609 (length logic :490-504, denoising loop :568-594 with `encoder_hidden_states=None,
    class_labels=prompt_embeds`), _encode_prompt :147-285 (CLAP text_embeds + F.normalize :217-219), check_inputs
    :320-378, prepare_latents :381-401, decode_latents :287-291, mel_spectrogram_to_waveform :293-300. Numerics on the
    HIP engines: ClapTextEngine, UNetEngine (class-embedding form), VAEDecoderEngine, HifiGanEngine."""

    def __init__(self, vae: VAEDecoderEngine, text_encoder: Optional[ClapTextEngine], tokenizer, unet: UNetEngine, scheduler,
                 vocoder: HifiGanEngine, sample_size: int = 128):
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.vocoder = vae, text_encoder, tokenizer, unet, vocoder
        self.scheduler = scheduler
        self.sample_size = sample_size
        self.device = unet.device
        self.vae_scale_factor = 2 ** (len(vae.cfg.block_out) - 1)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device="cuda:0", precise: int = 0, **unused):
        """diffusers layout: unet/, vae/, text_encoder/, tokenizer/, vocoder/, scheduler/scheduler_config.json."""
        from transformers import RobertaTokenizer
        sc = json.load(open(os.path.join(path, "scheduler", "scheduler_config.json")))
        sched = scheduler_from_config(sc)
        ucfg = json.load(open(os.path.join(path, "unet", "config.json")))
        dt = engine_dtype(torch_dtype)
        return cls(VAEDecoderEngine.from_pretrained(os.path.join(path, "vae"), device, dtype=dt),
                   ClapTextEngine.from_pretrained(os.path.join(path, "text_encoder"), device, dtype=dt),
                   RobertaTokenizer.from_pretrained(os.path.join(path, "tokenizer")),
                   UNetEngine.from_pretrained(os.path.join(path, "unet"), device, dtype=dt, stream32=True, precise=precise), sched,
                   HifiGanEngine.from_pretrained(os.path.join(path, "vocoder"), device, dtype=dt), ucfg.get("sample_size", 128))

    def to(self, device=None, *a, **k):
        return self

    # ------------------------------------------------------------------ prompt encoding (custom_ad.py:147-285)
    def _clap(self, text: List[str], max_length=None) -> torch.Tensor:
        tk = self.tokenizer
        enc = tk(text, padding="max_length", max_length=max_length or tk.model_max_length, truncation=True, return_tensors="pt")
        return self.text_encoder.text_embeds(enc.input_ids, enc.attention_mask, normalize=True)

    def _encode_prompt(self, prompt, num_waveforms_per_prompt, do_cfg, negative_prompt=None, prompt_embeds=None,
                       negative_prompt_embeds=None) -> torch.Tensor:
        if prompt is not None and isinstance(prompt, str):
            prompt = [prompt]
        batch_size = len(prompt) if prompt is not None else prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_embeds = self._clap(prompt)
        prompt_embeds = prompt_embeds.to(device=self.device, dtype=self.unet.dtype)
        bs, dim = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_waveforms_per_prompt).view(bs * num_waveforms_per_prompt, dim)
        if do_cfg and negative_prompt_embeds is None:
            if negative_prompt is None:
                uncond = [""] * batch_size
            elif prompt is not None and type(prompt) is not type(negative_prompt) and not isinstance(negative_prompt, str):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {type(prompt)}.")
            elif isinstance(negative_prompt, str):
                uncond = [negative_prompt]
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt` has batch size {len(negative_prompt)}, but `prompt` has batch size {batch_size}.")
            else:
                uncond = list(negative_prompt)
            # the reference pads the unconditional text to prompt_embeds.shape[1] (= the embedding width, custom_ad.py:255);
            # padding length does not change a masked encoder's output, so only the truncation limit matters
            negative_prompt_embeds = self._clap(uncond, max_length=min(prompt_embeds.shape[1], self.tokenizer.model_max_length))
        if do_cfg:
            n = negative_prompt_embeds.to(device=self.device, dtype=self.unet.dtype)
            n = n.repeat(1, num_waveforms_per_prompt).view(batch_size * num_waveforms_per_prompt, n.shape[1])
            prompt_embeds = torch.cat([n, prompt_embeds])   # [uncond | cond] (custom_ad.py:283)
        return prompt_embeds.contiguous()

    def check_inputs(self, prompt, audio_length_in_s, vocoder_upsample_factor, callback_steps, negative_prompt=None,
                     prompt_embeds=None, negative_prompt_embeds=None):
        min_len = vocoder_upsample_factor * self.vae_scale_factor
        if audio_length_in_s < min_len:
            raise ValueError(f"`audio_length_in_s` has to be a positive value greater than or equal to {min_len}, but is {audio_length_in_s}.")
        if self.vocoder.config.model_in_dim % self.vae_scale_factor != 0:
            raise ValueError(f"The number of frequency bins in the vocoder's log-mel spectrogram has to be divisible by the VAE scale "
                             f"factor, but got {self.vocoder.config.model_in_dim} bins and a scale factor of {self.vae_scale_factor}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly")

    def prepare_latents(self, batch_size, channels, height, generator, latents=None):
        shape = (batch_size, channels, height // self.vae_scale_factor, self.vocoder.config.model_in_dim // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch size of {batch_size}.")
        if latents is None:
            gdev = generator.device if generator is not None and not isinstance(generator, list) else self.device
            latents = torch.randn(shape, generator=generator if not isinstance(generator, list) else None, device=gdev,
                                  dtype=torch.float32).to(self.device)
        else:
            latents = latents.to(self.device, torch.float32)
        return (latents * self.scheduler.init_noise_sigma).contiguous()

    def decode_latents(self, latents) -> torch.Tensor:
        return self.vae.decode(latents, to_image=False)          # mel [B,1,frames,n_mel]; 1/scaling_factor folded into the VAE

    def mel_spectrogram_to_waveform(self, mel: torch.Tensor) -> torch.Tensor:
        if mel.dim() == 4:
            mel = mel.squeeze(1)
        return self.vocoder(mel).cpu().float()

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None] = None, audio_length_in_s: Optional[float] = None,
                 num_inference_steps: int = 10, guidance_scale: float = 2.5, negative_prompt=None,
                 num_waveforms_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, return_dict: bool = True, callback=None, callback_steps: int = 1,
                 cross_attention_kwargs=None, output_type: Optional[str] = "np", return_prompts_only: bool = False):
        vc = self.vocoder.config
        vocoder_upsample_factor = float(np.prod(vc.upsample_rates)) / vc.sampling_rate
        if audio_length_in_s is None:
            audio_length_in_s = self.sample_size * self.vae_scale_factor * vocoder_upsample_factor
        height = int(audio_length_in_s / vocoder_upsample_factor)
        original_waveform_length = int(audio_length_in_s * vc.sampling_rate)
        if height % self.vae_scale_factor != 0:
            height = int(np.ceil(height / self.vae_scale_factor)) * self.vae_scale_factor
        self.check_inputs(prompt, audio_length_in_s, vocoder_upsample_factor, callback_steps, negative_prompt, prompt_embeds,
                          negative_prompt_embeds)
        if isinstance(prompt, str):
            batch_size = 1
        elif isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0 and not return_prompts_only
        embeds = self._encode_prompt(prompt, num_waveforms_per_prompt, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds)
        if return_prompts_only:
            return embeds
        lat = self.prepare_latents(batch_size * num_waveforms_per_prompt, self.unet.cfg.in_ch, height, generator, latents)
        lat = denoise(self.unet, self.scheduler, lat, None, guidance_scale, num_inference_steps, class_labels=embeds)
        mel = self.decode_latents(lat)
        audio = self.mel_spectrogram_to_waveform(mel)[:, :original_waveform_length]
        if output_type == "np":
            audio = audio.numpy()
        return AudioPipelineOutput(audio) if return_dict else (audio,)


class TextToVideoSDPipelineOutput:
    def __init__(self, frames):
        self.frames = frames


def tensor2vid(video: torch.Tensor) -> List[np.ndarray]:
    """custom_vd.py:59-74. video [B,3,F,H,W] in [-1,1] -> F uint8 frames [H, B*W, 3] (the batch is tiled horizontally)."""
    video = (video * 0.5 + 0.5).clamp_(0, 1)
    i, c, f, h, w = video.shape
    images = video.permute(2, 3, 0, 4, 1).reshape(f, h, i * w, c)
    return [(im.cpu().numpy() * 255).astype("uint8") for im in images.unbind(0)]


@registry.register_model("vd")
class TextToVideoSDPipeline(StableDiffusionPipeline):
    """Text-to-video (zeroscope / modelscope) with the reference's call contract (spider/models/custom_vd.py):

        pipe = TextToVideoSDPipeline.from_pretrained(path, torch_dtype=...).to(device)
        pipe(prompt=[...] | prompt_embeds=T[B,77,C], height, width, num_frames, num_inference_steps, guidance_scale,
             negative_prompt, generator, latents, output_type="np"|"pt", return_prompts_only=False) -> .frames

    __call__ :506-716 (denoising loop :664-697), _encode_prompt :223-379 (plain right truncation, unlike custom_sd),
    prepare_latents :483-503, decode_latents :381-408, tensor2vid :59-74. Numerics on the HIP engines: CLIPTextEngine,
    UNet3DEngine, VAEDecoderEngine (frames decoded as a batch of images)."""

    def __init__(self, unet: UNet3DEngine, vae: Optional[VAEDecoderEngine], text_encoder: Optional[CLIPTextEngine], tokenizer,
                 scheduler=None, sample_size: int = 32):
        from .schedulers import DDIMScheduler
        super().__init__(unet, vae, text_encoder, tokenizer, scheduler or DDIMScheduler(), sample_size)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device="cuda:0", precise: int = 0, **unused):
        from transformers import CLIPTokenizer
        sc = json.load(open(os.path.join(path, "scheduler", "scheduler_config.json")))
        sched = scheduler_from_config(sc)
        ucfg = json.load(open(os.path.join(path, "unet", "config.json")))
        dt = engine_dtype(torch_dtype)
        return cls(UNet3DEngine.from_pretrained(os.path.join(path, "unet"), device, dtype=dt, stream32=True, precise=precise),
                   VAEDecoderEngine.from_pretrained(os.path.join(path, "vae"), device, dtype=dt),       # scaling_factor from the config (:382)
                   CLIPTextEngine.from_pretrained(os.path.join(path, "text_encoder"), device, dtype=dt),
                   CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer")), sched, ucfg.get("sample_size", 32))

    def _tokenize(self, prompt: List[str]) -> torch.Tensor:
        tk = self.tokenizer
        return tk(prompt, padding="max_length", max_length=tk.model_max_length, truncation=True, return_tensors="pt").input_ids

    def prepare_latents(self, batch_size, channels, num_frames, height, width, generator, latents=None):
        shape = (batch_size, channels, num_frames, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch size of {batch_size}.")
        if latents is None:
            gdev = generator.device if generator is not None and not isinstance(generator, list) else self.device
            latents = torch.randn(shape, generator=generator if not isinstance(generator, list) else None, device=gdev,
                                  dtype=torch.float32).to(self.device)
        else:
            latents = latents.to(self.device, torch.float32)
        return (latents * self.scheduler.init_noise_sigma).contiguous()

    def decode_latents(self, latents) -> torch.Tensor:
        """[B,4,F,h,w] -> video [B,3,F,8h,8w] fp32 in [-1,1] (custom_vd.py:381-408)."""
        B, C, F_, h, w = latents.shape
        flat = latents.permute(0, 2, 1, 3, 4).reshape(B * F_, C, h, w).contiguous()
        frames = torch.cat([self.vae.decode(flat[i:i + 4], to_image=False) for i in range(0, B * F_, 4)])   # bounded VAE batches
        return frames.view(B, F_, *frames.shape[1:]).permute(0, 2, 1, 3, 4).float()

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_frames: int = 16, num_inference_steps: int = 50, guidance_scale: float = 9.0, negative_prompt=None,
                 eta: float = 0.0, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 output_type: Optional[str] = "np", return_dict: bool = True, callback=None, callback_steps: int = 1,
                 cross_attention_kwargs=None, return_prompts_only: bool = False):
        height = height or self.sample_size * self.vae_scale_factor
        width = width or self.sample_size * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0 and not return_prompts_only
        embeds = self._encode_prompt(prompt, 1, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds)
        if return_prompts_only:
            return embeds
        lat = self.prepare_latents(batch_size, self.unet.cfg.in_ch, num_frames, height, width, generator, latents)
        lat = video_denoise(self.unet, self.scheduler, lat, embeds, guidance_scale, num_inference_steps)
        video = self.decode_latents(lat)
        if output_type != "pt":
            video = tensor2vid(video)
        return TextToVideoSDPipelineOutput(video) if return_dict else (video,)
