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


@pytest.mark.parametrize("depth", [2, 3])
def test_pipelined_soak_changing_geometry(dev, depth):
    """28 chat requests whose prompt lengths change in runs (every change is a new LLM geometry: its first pass must NOT be
    overlapped -- it captures graphs), some answered twice in a row: the pipelined results equal `predict` of the same requests
    bit for bit (response text, token ids, image bytes), arrive in request order, and the device memory in use stops growing once
    every geometry has been seen."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    bench = _bench()
    resp = bench.Responder(_args(prompt_len=96, new_tokens=8, denoise_steps=3, pipeline_depth=depth), dev)
    infer = resp.infer
    words = [12, 12, 12, 30, 30, 12, 12, 50, 50, 50, 12, 30, 30, 30, 50, 12, 12, 12, 12, 30, 50, 50, 12, 12, 30, 30, 12, 12]
    reqs = [[{"role": "user", "content": " ".join(f"w{(7 * i + j) % 91}" for j in range(n))}] for i, n in enumerate(words)]
    import numpy as np

    def key(r):
        return r.response, r.text_ids.cpu(), np.asarray(r.predictions["IMAGE"][0]).copy()

    torch.cuda.manual_seed(77)
    ref = [key(infer(m)) for m in reqs]
    torch.cuda.synchronize()
    torch.cuda.manual_seed(77)
    got, mem = [], []
    for r in infer.pipelined(reqs):
        got.append(key(r))
        mem.append(torch.cuda.memory_allocated(dev))
    assert len(got) == len(reqs)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g[0] == r[0], f"request {i}: response text differs"
        assert torch.equal(g[1], r[1]), f"request {i}: token ids differ"
        assert np.array_equal(g[2], r[2]), f"request {i}: image differs"
    assert len({g[0] for g in got}) > 3, "the requests are different requests"
    # all three geometries were seen by request 10: no GROWTH afterwards beyond one allocator block (memory released by the garbage
    # collection of earlier tests' engines in the middle of the run is not this test's business)
    assert max(mem[12:]) - mem[12] <= 64 << 20, [m >> 20 for m in mem]
    # and a second sweep over the same requests (everything warm: every pass overlapped) is still identical
    torch.cuda.manual_seed(77)
    again = [key(r) for r in infer.pipelined(reqs)]
    assert all(a[0] == r[0] and torch.equal(a[1], r[1]) and np.array_equal(a[2], r[2]) for a, r in zip(again, ref))


def test_pipelined_groups_answer_conversations_together(dev):
    """`pipelined(conversations, group=2)` on the real engines: pairs of conversations of different lengths go through ONE batched LLM
    pass (left-padded rows) and ONE batched decoder pass, overlapped with the neighbouring pairs; every conversation gets the result
    the same pair gets from `predict` on the batched request (token ids and image bytes equal), in request order."""
    import numpy as np
    bench = _bench()
    resp = bench.Responder(_args(batch=2, prompt_len=96, new_tokens=8, denoise_steps=3), dev)
    infer = resp.infer
    words = [12, 30, 30, 12, 20, 20, 41, 12, 12]
    convs = [[{"role": "user", "content": " ".join(f"w{(5 * i + j) % 83}" for j in range(n))}] for i, n in enumerate(words)]
    key = lambda r: (r.response, r.text_ids.cpu(), np.asarray(r.predictions["IMAGE"][0]).copy())
    torch.cuda.manual_seed(5)
    ref = []
    for i in range(0, len(convs), 2):
        out = infer.predict(inputs=infer.build_inputs_batch(convs[i:i + 2]))
        ref += [key(r) for r in (out if isinstance(out, list) else [out])]
    torch.cuda.manual_seed(5)
    got = [key(r) for r in infer.pipelined(convs, group=2)]
    assert len(got) == len(convs) == len(ref)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g[0] == r[0] and torch.equal(g[1], r[1]) and np.array_equal(g[2], r[2]), f"conversation {i}"
    assert len({g[0] for g in got}) > 3
