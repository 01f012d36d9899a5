"""CPU: which weight files a `from_pretrained` reads from a published component directory (spider_amd/checkpoint.py) -- the
layouts behind the reference's loader calls (base_model.py:207-219, spider_decoder.py:109,114, r1_llama3_8B_infer.py:4)."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

from spider_amd.checkpoint import load_state_dict, read_config, resolve_weight_files


def _w(v):
    return {"a.weight": torch.full((2, 3), float(v)), "b.bias": torch.full((3,), float(v))}


def test_published_diffusers_directory_reads_exactly_the_plain_safetensors(tmp_path):
    """runwayml/stable-diffusion-v1-5/unet holds six weight files side by side; one of them is the model."""
    d = str(tmp_path)
    save_file(_w(1), os.path.join(d, "diffusion_pytorch_model.safetensors"))
    save_file(_w(2), os.path.join(d, "diffusion_pytorch_model.fp16.safetensors"))
    save_file(_w(3), os.path.join(d, "diffusion_pytorch_model.non_ema.safetensors"))
    torch.save(_w(4), os.path.join(d, "diffusion_pytorch_model.bin"))
    torch.save(_w(5), os.path.join(d, "diffusion_pytorch_model.fp16.bin"))
    assert [os.path.basename(f) for f in resolve_weight_files(d)] == ["diffusion_pytorch_model.safetensors"]
    assert load_state_dict(d)["a.weight"][0, 0] == 1
    assert load_state_dict(d, variant="fp16")["a.weight"][0, 0] == 2
    assert load_state_dict(d, variant="non_ema")["b.bias"][0] == 3
    with pytest.raises(FileNotFoundError):
        resolve_weight_files(d, variant="bf16")


def test_bin_only_directory(tmp_path):
    """cerspense/zeroscope_v2_576w: torch pickles only."""
    d = str(tmp_path)
    torch.save(_w(7), os.path.join(d, "diffusion_pytorch_model.bin"))
    sd = load_state_dict(d)
    assert set(sd) == {"a.weight", "b.bias"} and sd["a.weight"][1, 2] == 7
    os.remove(os.path.join(d, "diffusion_pytorch_model.bin"))
    torch.save(_w(8), os.path.join(d, "pytorch_model.bin"))       # transformers text encoder
    assert load_state_dict(d)["b.bias"][1] == 8


def test_only_a_variant_present_is_taken_with_a_warning(tmp_path):
    d = str(tmp_path)
    save_file(_w(2), os.path.join(d, "model.fp16.safetensors"))
    with pytest.warns(UserWarning, match="fp16"):
        assert load_state_dict(d)["a.weight"][0, 0] == 2


def test_sharded_index_and_keep_filter(tmp_path):
    """An LLM directory: model.safetensors.index.json + shards; a stray consolidated file beside them is not read."""
    d = str(tmp_path)
    save_file({"model.x": torch.ones(2), "visual.y": torch.ones(3)}, os.path.join(d, "model-00001-of-00002.safetensors"))
    save_file({"lm_head.z": torch.ones(4) * 2}, os.path.join(d, "model-00002-of-00002.safetensors"))
    save_file({"model.x": torch.zeros(2)}, os.path.join(d, "zz_consolidated.safetensors"))
    json.dump({"metadata": {}, "weight_map": {"model.x": "model-00001-of-00002.safetensors", "visual.y": "model-00001-of-00002.safetensors",
                                              "lm_head.z": "model-00002-of-00002.safetensors"}},
              open(os.path.join(d, "model.safetensors.index.json"), "w"))
    assert [os.path.basename(f) for f in resolve_weight_files(d)] == ["model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors"]
    sd = load_state_dict(d, keep=lambda k: k if k.startswith(("model.", "lm_head.")) else None)
    assert set(sd) == {"model.x", "lm_head.z"} and sd["model.x"][0] == 1
    os.remove(os.path.join(d, "model-00002-of-00002.safetensors"))
    with pytest.raises(FileNotFoundError, match="shard"):
        resolve_weight_files(d)


def test_shards_without_an_index_and_empty_directories(tmp_path):
    d = str(tmp_path)
    with pytest.raises(FileNotFoundError, match="no .* weight file"):
        resolve_weight_files(d)
    with pytest.raises(FileNotFoundError):
        resolve_weight_files(os.path.join(d, "missing"))
    with pytest.raises(FileNotFoundError, match="config.json"):
        read_config(d)
    save_file({"p": torch.ones(1)}, os.path.join(d, "part-b.safetensors"))
    save_file({"q": torch.ones(1)}, os.path.join(d, "part-a.safetensors"))
    assert [os.path.basename(f) for f in resolve_weight_files(d)] == ["part-a.safetensors", "part-b.safetensors"]
    assert set(load_state_dict(d)) == {"p", "q"}
    with pytest.raises(ValueError, match="no tensor"):
        load_state_dict(d, keep=lambda k: None)


def test_pickle_loading_is_weights_only(tmp_path):
    """A .bin is an untrusted pickle: anything but tensors and containers is refused (torch.load(weights_only=True))."""
    import pickle
    d = str(tmp_path)

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    torch.save({"a": Evil()}, os.path.join(d, "pytorch_model.bin"))
    with pytest.raises((pickle.UnpicklingError, RuntimeError)):
        load_state_dict(d)
