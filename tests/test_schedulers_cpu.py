"""CPU: host-side scheduler coefficient logic (spider_amd/schedulers.py) against the oracle's scheduler
restatement -- timesteps identical, and the per-step linear-combination coefficients reproduce the oracle's
update when applied with plain tensor math (the device lincomb kernel itself is covered by the GPU tests)."""
import torch

from oracle.unet import DDIMOracle, PNDMOracle
from spider_amd import schedulers as S


def _run(prod, orc, steps, monkey):
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 8, 8, generator=g)
    ts_p, ts_o = prod.set_timesteps(steps), orc.set_timesteps(steps)
    assert torch.equal(ts_p, ts_o)
    xp, xo = x0.clone(), x0.clone()
    for t in ts_o:
        e = torch.randn(1, 4, 8, 8, generator=g)
        xp = prod.step(e, t, xp)
        xo = orc.step(e, t, xo)
        assert torch.allclose(xp, xo, atol=1e-5, rtol=1e-5)
    return ts_p


def test_schedulers_match_oracle(monkeypatch):
    monkeypatch.setattr(S.ops, "lincomb", lambda ts, cs, out=None: sum(c * t for c, t in zip(cs, ts)))
    ts = _run(S.PNDMScheduler(), PNDMOracle(), 40, monkeypatch)
    assert len(ts) == 41 and int(ts[0]) == 976 and int(ts[-1]) == 1   # 40 steps -> 41 UNet calls
    assert int(ts[1]) == int(ts[2]) == 951                              # the repeated PLMS warm-up timestep
    ts = _run(S.DDIMScheduler(), DDIMOracle(), 50, monkeypatch)
    assert len(ts) == 50 and int(ts[0]) == 981 and int(ts[-1]) == 1


def test_scheduler_config_is_honoured_or_refused():
    import pytest
    sd15 = {"_class_name": "PNDMScheduler", "_diffusers_version": "0.6.0", "beta_end": 0.012, "beta_schedule": "scaled_linear",
            "beta_start": 0.00085, "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True,
            "steps_offset": 1, "trained_betas": None, "clip_sample": False}
    s = S.scheduler_from_config(sd15)
    assert isinstance(s, S.PNDMScheduler) and float(s.final_alpha) == float(s.ac[0])
    s1 = S.scheduler_from_config({**sd15, "_class_name": "DDIMScheduler", "set_alpha_to_one": True})
    assert float(s1.final_alpha) == 1.0
    lin = S.scheduler_from_config({**sd15, "beta_schedule": "linear"})
    assert not torch.allclose(lin.ac, s.ac)
    for bad in ({"prediction_type": "v_prediction"}, {"timestep_spacing": "trailing"}, {"beta_schedule": "squaredcos_cap_v2"},
                {"skip_prk_steps": False}, {"_class_name": "EulerDiscreteScheduler"}):
        with pytest.raises(NotImplementedError):
            S.scheduler_from_config({**sd15, **bad})
    with pytest.raises(NotImplementedError):
        S.scheduler_from_config({**sd15, "_class_name": "DDIMScheduler", "clip_sample": True})
    # a key the JSON omits takes the diffusers-0.25 constructor default, not this module's: DDIM set_alpha_to_one=True and
    # clip_sample=True (refused), PNDM skip_prk_steps=False (refused); rescale_betas_zero_snr changes the betas (refused)
    ddim_min = {"_class_name": "DDIMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear", "clip_sample": False}
    d = S.scheduler_from_config(ddim_min)
    assert float(d.final_alpha) == 1.0 and d.offset == 0
    with pytest.raises(NotImplementedError, match="clip_sample"):
        S.scheduler_from_config({k: v for k, v in ddim_min.items() if k != "clip_sample"})
    with pytest.raises(NotImplementedError, match="skip_prk_steps"):
        S.scheduler_from_config({"_class_name": "PNDMScheduler"})
    with pytest.raises(NotImplementedError, match="rescale_betas_zero_snr"):
        S.scheduler_from_config({**ddim_min, "rescale_betas_zero_snr": True})
