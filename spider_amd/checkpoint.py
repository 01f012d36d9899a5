"""Weight-file resolution for every `from_pretrained` of the package: the on-disk layouts the reference's loaders read.

The reference loads its decoders with `cls.from_pretrained(ckpt, torch_dtype=torch.float16)` (base_model.py:207-219,
spider_decoder.py:109,114), its LLMs with `AutoModelForCausalLM.from_pretrained` (r1_llama3_8B_infer.py:4) and
`Qwen2_5OmniModel.from_pretrained` (qwen2.5omni_spider_web.py:376-381). What those calls accept in a component directory, in
their order of preference, is what `resolve_weight_files` accepts:

  1. `<stem>.safetensors.index.json`  -> the shard files its `weight_map` names
  2. `<stem>.safetensors`
  3. `<stem>.bin.index.json` / `<stem>.bin`   (torch pickles: zeroscope_v2_576w ships only these), read with `weights_only=True`

with `<stem>` one of `diffusion_pytorch_model` (diffusers models), `model` / `pytorch_model` (transformers models). A published
diffusers directory holds several files side by side (`diffusion_pytorch_model.safetensors`, `….fp16.safetensors`,
`….non_ema.safetensors`, the same as `.bin`): exactly one weight set is read -- the plain one, or `variant="fp16"` etc. when asked;
when only a variant exists it is taken with a warning. Directories with other file names (e.g. shards without an index) fall back
to every `*.safetensors`, then every `*.bin`, in sorted order. Nothing found is a FileNotFoundError that lists the directory.
"""
import glob
import json
import os
import warnings
from typing import Callable, Dict, List, Optional

import torch

STEMS = ("diffusion_pytorch_model", "model", "pytorch_model")
_KNOWN_VARIANTS = ("fp16", "non_ema", "ema_only", "bf16")


def _index_files(path: str, stem: str, ext: str, variant: Optional[str]) -> List[str]:
    """Shard list of `<stem>.<ext>.index.json` (variant forms: `<stem>.<ext>.index.<v>.json`, `<stem>.<v>.<ext>.index.json`)."""
    names = [f"{stem}.{ext}.index.json"] if variant is None else [f"{stem}.{ext}.index.{variant}.json", f"{stem}.{variant}.{ext}.index.json"]
    for n in names:
        p = os.path.join(path, n)
        if os.path.exists(p):
            wm = json.load(open(p)).get("weight_map", {})
            files = sorted({os.path.join(path, f) for f in wm.values()})
            missing = [f for f in files if not os.path.exists(f)]
            if missing:
                raise FileNotFoundError(f"{p} names shard files that do not exist: {[os.path.basename(m) for m in missing]}")
            if files:
                return files
    return []


def _single(path: str, stem: str, ext: str, variant: Optional[str]) -> List[str]:
    p = os.path.join(path, f"{stem}.{ext}" if variant is None else f"{stem}.{variant}.{ext}")
    return [p] if os.path.exists(p) else []


def _resolve(path: str, variant: Optional[str]) -> List[str]:
    for ext in ("safetensors", "bin"):
        for stem in STEMS:
            files = _index_files(path, stem, ext, variant) or _single(path, stem, ext, variant)
            if files:
                return files
    return []


def resolve_weight_files(path: str, variant: Optional[str] = None) -> List[str]:
    """The ONE set of weight files a component directory is loaded from (see the module docstring for the order)."""
    if not os.path.isdir(path):
        raise FileNotFoundError(f"checkpoint directory {path!r} does not exist")
    files = _resolve(path, variant)
    if files:
        return files
    if variant is not None:
        raise FileNotFoundError(f"no weight files of variant {variant!r} in {path!r} (found: {sorted(os.listdir(path))})")
    for v in _KNOWN_VARIANTS:       # only a variant was published / downloaded
        files = _resolve(path, v)
        if files:
            warnings.warn(f"{path}: no plain weight file, loading the {v!r} variant ({os.path.basename(files[0])})")
            return files
    for ext in ("safetensors", "bin"):      # other naming (shards without an index, single-file exports)
        files = sorted(glob.glob(os.path.join(path, f"*.{ext}")))
        if files:
            return files
    raise FileNotFoundError(f"no *.safetensors / *.bin weight file in {path!r} (found: {sorted(os.listdir(path))})")


def load_state_dict(path: str, variant: Optional[str] = None,
                    keep: Optional[Callable[[str], Optional[str]]] = None) -> Dict[str, torch.Tensor]:
    """name -> CPU tensor of a component directory. `keep(name)` returns the name to store the tensor under, or None to skip
    it WITHOUT reading it (safetensors) -- an LLM engine loading from a Qwen2.5-Omni checkpoint never touches the towers and
    the talker."""
    out: Dict[str, torch.Tensor] = {}
    for f in resolve_weight_files(path, variant):
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(f, framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    kk = keep(k) if keep is not None else k
                    if kk is not None:
                        out[kk] = sf.get_tensor(k)
        else:
            sd = torch.load(f, map_location="cpu", weights_only=True)
            if isinstance(sd, dict) and isinstance(sd.get("state_dict"), dict):       # trainer-style wrapper around the tensors
                sd = sd["state_dict"]
            for k, v in sd.items():
                if not torch.is_tensor(v):
                    continue
                kk = keep(k) if keep is not None else k
                if kk is not None:
                    out[kk] = v
    if not out:
        raise ValueError(f"{path!r}: the weight files hold no tensor this engine reads")
    return out


def read_config(path: str, name: str = "config.json") -> dict:
    p = os.path.join(path, name)
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found: {path!r} is not a diffusers / transformers component directory")
    with open(p) as fh:
        return json.load(fh)
