"""GPU: the two-stream pipelining of `spider_amd.SpiderFreeInfer` (the decoder pass of request k beside the LLM pass of request k+1,
each in its own split-K workspace scope, the decoder pass enqueued by a helper thread) returns bit-identical results to the
one-stream order `SpiderFreeInfer.predict`: the two passes share no mutable state. Driven through bench.py's thin Responder, i.e.
through the whole product chain QwenOmniThinker.generate -> batch_decode -> extract_answer -> SpiderDecoderInfer ->
SpiderDecoder.generate -> StableDiffusionPipeline."""
import argparse
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.DIFF_DT = torch.float16
    return bench


def _args(**kw):
    base = dict(llm="qwen25_7b", batch=1, throughput_batch=0, prompt_len=192, new_tokens=12, denoise_steps=5, schedule="overlap",
                workload="text_image", no_stream32=False, serial_decoders=False, pipeline_depth=2)
    base.update(kw)
    return argparse.Namespace(**base)


@pytest.mark.parametrize("depth", [2, 3])
def test_overlapped_schedule_equals_serial(dev, depth):
    """depth 3: the prompt pass of request k+2 rides on the decoder stream behind the decoder pass of request k while request k+1
    decodes from the other KV cache set"""
    bench = _bench()
    resp = bench.Responder(_args(pipeline_depth=depth), dev)
    assert resp.infer.depth == depth
    from spider_amd import SpiderFreeInfer
    assert isinstance(resp.infer, SpiderFreeInfer)
    torch.cuda.manual_seed(1234)                       # the pipeline draws its latents from the device's default generator
    ref = [{k: v.clone() for k, v in resp.respond_serial().items()} for _ in range(4)]
    torch.cuda.manual_seed(1234)
    outs = [resp.respond() for _ in range(4)]          # call 1 primes the pipeline (passes run alone), the later calls overlap the passes
    last = resp.infer.flush()
    torch.cuda.synchronize()
    assert resp.overlap_ms and resp.overlap_ms["decoder_pass_ms"] > 0 and resp.overlap_ms["llm_pass_ms"] > 0
    for r, o in zip(ref, outs):
        assert torch.equal(o["tokens"], r["tokens"]), "tokens differ between the pipelined and the serial schedule"
        assert torch.equal(o["out"], r["out"]), "image differs between the pipelined and the serial schedule"
    assert last is not None and last.response == resp.processor.batch_decode(last.text_ids[None])[0].split("\n")[-1]
    img = ref[0]["out"]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (1, 3, 512, 512) and int(img.max()) > int(img.min())
    assert not torch.equal(ref[0]["out"], ref[1]["out"]), "consecutive requests draw different latents"


def test_spider_free_infer_contract(dev):
    """messages in, (response, answers, predictions, predictions_text) out, like `predict` (qwen2.5omni_spider_web.py:458-521);
    a batch of rows goes through generate_batch; `pipelined()` yields every request's result in order."""
    bench = _bench()
    resp = bench.Responder(_args(batch=2, prompt_len=64, new_tokens=6, denoise_steps=2), dev)
    infer = resp.infer
    res = infer([{"role": "user", "content": "draw a red apple"}])
    assert res.response.startswith("Sure. <IMAGE>scene ") and res.answers == [res.response]
    assert res.predictions_text["IMAGE"] == [res.response[len("Sure. <IMAGE>"):-len("</IMAGE>")]]
    assert len(res.predictions["IMAGE"]) == 1 and res.predictions["IMAGE"][0].size == (512, 512)
    a, p, pt = res                                      # unpacks like spider_decoder_infer(ask_info)
    assert a is res.answers and p is res.predictions and pt is res.predictions_text
    two = infer(inputs=resp.request(2))
    assert isinstance(two, list) and len(two) == 2 and all(len(r.predictions["IMAGE"]) == 1 for r in two)
    reqs = [resp.request(1) for _ in range(4)]
    got = list(infer.pipelined(reqs))
    assert len(got) == 4 and all(r.response == got[0].response for r in got)
    with pytest.raises(ValueError):
        infer()
