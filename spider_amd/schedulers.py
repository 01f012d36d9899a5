"""Scheduler coefficient logic (host side) for the denoising loop, custom_sd.py:608,647.

The reference takes its scheduler class from the checkpoint (diffusers==0.25.0): PNDMScheduler with
skip_prk_steps for SD-v1.5 (40 steps -> 41 UNet calls), DDIMScheduler for StoryDiffusion
(Comic_Generation.py:316-317). Every `scheduler.step` of both is a linear combination of the current sample
and (stored) noise predictions; this module only produces the coefficients, the update itself is ONE
spider_lincomb_f32 launch on fp32 latents in HBM.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import ops


def _alphas_cumprod(beta_start=0.00085, beta_end=0.012, n=1000, beta_schedule="scaled_linear") -> torch.Tensor:
    if beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    elif beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    else:
        raise NotImplementedError(f"beta_schedule {beta_schedule!r} (scaled_linear and linear are implemented)")
    return torch.cumprod(1.0 - betas, 0)


def _check_config(name: str, prediction_type="epsilon", timestep_spacing="leading", clip_sample=False, thresholding=False,
                  trained_betas=None, skip_prk_steps=True, rescale_betas_zero_snr=False, **ignored):
    """The fields of scheduler_config.json whose non-default values change the update rule: refuse them instead of
    silently denoising with the wrong coefficients (the checkpoints on the reference path -- SD-v1.5 PNDM, SDXL /
    zeroscope / AudioLDM DDIM -- all carry the values accepted here)."""
    bad = []
    if prediction_type != "epsilon":
        bad.append(f"prediction_type={prediction_type!r}")
    if timestep_spacing != "leading":
        bad.append(f"timestep_spacing={timestep_spacing!r}")
    if clip_sample and name == "DDIMScheduler":
        bad.append("clip_sample=True")
    if thresholding:
        bad.append("thresholding=True")
    if trained_betas is not None:
        bad.append("trained_betas")
    if rescale_betas_zero_snr:
        bad.append("rescale_betas_zero_snr=True")
    if name == "PNDMScheduler" and not skip_prk_steps:
        bad.append("skip_prk_steps=False")
    if bad:
        raise NotImplementedError(f"{name}: unsupported scheduler configuration: " + ", ".join(bad))


class PNDMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 beta_schedule="scaled_linear", set_alpha_to_one=False, **cfg):
        _check_config("PNDMScheduler", **cfg)
        self.ac = _alphas_cumprod(beta_start, beta_end, num_train_timesteps, beta_schedule)
        self.final_alpha = torch.tensor(1.0) if set_alpha_to_one else self.ac[0]
        self.n_train, self.offset = num_train_timesteps, steps_offset

    def set_timesteps(self, n: int):
        self.n = n
        ratio = self.n_train // n
        ts = (torch.arange(0, n) * ratio).round().long() + self.offset
        self.timesteps = torch.cat([ts[:-1], ts[-2:-1], ts[-1:]]).flip(0)
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_sample = None
        return self.timesteps

    def scale_model_input(self, x, t):
        return x

    def _coeffs(self, t: int, prev_t: int) -> Tuple[float, float]:
        a_t = self.ac[t]
        a_p = self.ac[prev_t] if prev_t >= 0 else self.final_alpha
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return float(sample_coeff), float((a_p - a_t) / denom)

    def step(self, eps: torch.Tensor, t, sample: torch.Tensor) -> torch.Tensor:
        """eps, sample: fp32 device tensors. Returns the previous sample (new tensor)."""
        t = int(t)
        ratio = self.n_train // self.n
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_t, t = t, t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            terms, self.cur_sample = [(1.0, eps)], sample
        elif len(self.ets) == 1 and self.counter == 1:
            terms = [(0.5, eps), (0.5, self.ets[-1])]
            sample, self.cur_sample = self.cur_sample, None
        elif len(self.ets) == 2:
            terms = [(1.5, self.ets[-1]), (-0.5, self.ets[-2])]
        elif len(self.ets) == 3:
            terms = [(23 / 12, self.ets[-1]), (-16 / 12, self.ets[-2]), (5 / 12, self.ets[-3])]
        else:
            terms = [(55 / 24, self.ets[-1]), (-59 / 24, self.ets[-2]), (37 / 24, self.ets[-3]), (-9 / 24, self.ets[-4])]
        cs, cm = self._coeffs(t, prev_t)
        self.counter += 1
        return ops.lincomb([sample] + [e for _, e in terms], [cs] + [-cm * c for c, _ in terms])


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 beta_schedule="scaled_linear", set_alpha_to_one=False, **cfg):
        _check_config("DDIMScheduler", **cfg)
        self.ac = _alphas_cumprod(beta_start, beta_end, num_train_timesteps, beta_schedule)
        self.final_alpha = torch.tensor(1.0) if set_alpha_to_one else self.ac[0]
        self.n_train, self.offset = num_train_timesteps, steps_offset

    def set_timesteps(self, n: int):
        self.n = n
        ratio = self.n_train // n
        self.timesteps = (torch.arange(0, n) * ratio).round().flip(0).long() + self.offset
        return self.timesteps

    def scale_model_input(self, x, t):
        return x

    def step(self, eps: torch.Tensor, t, sample: torch.Tensor) -> torch.Tensor:
        t = int(t)
        prev_t = t - self.n_train // self.n
        a_t = self.ac[t]
        a_p = self.ac[prev_t] if prev_t >= 0 else self.final_alpha
        cx = float((a_p / a_t) ** 0.5)
        ce = float((1 - a_p) ** 0.5 - (a_p / a_t) ** 0.5 * (1 - a_t) ** 0.5)
        return ops.lincomb([sample, eps], [cx, ce])


SCHEDULERS = {"PNDMScheduler": PNDMScheduler, "DDIMScheduler": DDIMScheduler}

# diffusers==0.25.0 constructor defaults: what a scheduler_config.json that OMITS a key means (the Python constructors above default
# to the values of the checkpoints on the reference path instead, for direct construction in tests / bench)
_DIFFUSERS_DEFAULTS = {
    "DDIMScheduler": dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", steps_offset=0,
                          set_alpha_to_one=True, clip_sample=True),
    "PNDMScheduler": dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", steps_offset=0,
                          set_alpha_to_one=False, skip_prk_steps=False),
}


def scheduler_from_config(sc: dict):
    """scheduler/scheduler_config.json -> scheduler object. Unknown classes and update-rule options that are not
    implemented raise (no silent default)."""
    name = sc.get("_class_name")
    if name not in SCHEDULERS:
        raise NotImplementedError(f"scheduler class {name!r} (implemented: {sorted(SCHEDULERS)})")
    cfg = dict(_DIFFUSERS_DEFAULTS[name])
    cfg.update({k: v for k, v in sc.items() if not k.startswith("_")})
    return SCHEDULERS[name](**cfg)
