"""CPU: host logic of spider_amd.SpiderFreeInfer (the `predict` flow of qwen2.5omni_spider_web.py:458-521) with stand-in engines:
call order, the last-line / extract_answer rules, ask_info keys, batched rows, and the request pipelining state machine (results in
request order, every request exactly one LLM pass and one decoder pass, passes of a new geometry never overlapped)."""
import threading

import pytest
import torch

from spider_amd import SpiderDecoderInfer, SpiderFreeInfer
from benchkit.synthetic import SyntheticOmniProcessor


class FakeThinker:
    def __init__(self):
        self.calls = []

    def generate(self, input_ids, attention_mask=None, **kw):
        self.calls.append((threading.current_thread().name, tuple(input_ids.shape), dict(kw)))
        new = (input_ids[:, :1] + torch.arange(11, 15)[None]) % 97      # (never 1: the synthetic processor's assistant marker)
        return torch.cat([input_ids, new], 1)


class FakeThinker3(FakeThinker):
    """with the split API of QwenOmniThinker: prefill_begin / decode_finish (and the KV cache set each request lives in)"""

    def __init__(self):
        super().__init__()
        self.pre, self.dec = [], []

    def prefill_begin(self, input_ids, attention_mask=None, cache_set=0, **kw):
        self.pre.append((threading.current_thread().name, cache_set, int(input_ids[0, 0])))
        return ("handle", input_ids, cache_set)

    def adopt(self, handle, cache_set=0):
        self.adopted = getattr(self, "adopted", 0) + 1
        return (handle[0], handle[1], cache_set)

    def decode_finish(self, handle):
        _, input_ids, cache_set = handle
        self.dec.append((threading.current_thread().name, cache_set, int(input_ids[0, 0])))
        return FakeThinker.generate(self, input_ids)


class FakePipe:
    """StableDiffusionPipeline surface as SpiderDecoder uses it: pipe(prompt=[...], **kwargs).images"""
    def __init__(self):
        self.calls = []

    def __call__(self, prompt=None, **kw):
        self.calls.append((threading.current_thread().name, list(prompt), kw))
        class O:
            images = [f"img:{p}" for p in prompt]
        return O()


def make(tags=("IMAGE",), **kw):
    pipe, thinker = FakePipe(), FakeThinker()
    dinf = SpiderDecoderInfer({"model": dict(type="spider_decoder", pipelines={"IMAGE": pipe}, device="cpu",
                                             decode_kwargs={"IMAGE": {"num_inference_steps": 7}})})
    proc = SyntheticOmniProcessor(vocab=500, tags=tags, head=3)
    return SpiderFreeInfer(thinker, proc, dinf, device="cpu", generate_kwargs=dict(spk="Chelsie", use_audio_in_video=True), **kw), pipe, thinker


def test_predict_flow_matches_reference_order():
    infer, pipe, thinker = make()
    res = infer([{"role": "user", "content": "draw a cat"}])
    S = infer.processor.prompt_len
    head = " ".join(str(int(t)) for t in res.text_ids[S:S + 3])
    assert res.response == f"Sure. <IMAGE>scene {head}</IMAGE>"              # last line of the decoded text (:471)
    assert res.answers == [res.response] and res.predictions_text["IMAGE"] == [f"scene {head}"]
    assert res.predictions["IMAGE"] == [f"img:scene {head}"]
    assert thinker.calls[0][2] == {"spk": "Chelsie", "use_audio_in_video": True}   # generate(**inputs, spk=..., use_audio_in_video=True) (:468)
    assert pipe.calls[0][2]["num_inference_steps"] == 7 and pipe.calls[0][2]["guidance_scale"] == 7.5
    a, p, pt = res
    assert (a, p, pt) == (res.answers, res.predictions, res.predictions_text)


def test_extract_answer_and_image_inputs():
    infer, pipe, _ = make()

    class Proc(SyntheticOmniProcessor):
        def batch_decode(self, ids, **kw):
            return ["user\nassistant\n<think><IMAGE>not this</IMAGE></think>ok <IMAGE>this one</IMAGE>"]
    infer.processor = Proc(vocab=500)
    infer.process_mm_info = lambda messages, use_audio_in_video: (None, [[[1, 2, 3]]], None)
    seen = {}
    orig = infer.spider_decoder_infer.spider_decoder.generate

    def spy(samples, *a):
        seen.update(samples)
        return orig(samples, *a)
    infer.spider_decoder_infer.spider_decoder.generate = spy
    res = infer([{"role": "user", "content": "x"}])
    assert res.predictions_text["IMAGE"] == ["this one"]                   # only the text after </think> is routed (:341-347)
    assert seen["llm_text_all"] == ["ok <IMAGE>this one</IMAGE>"] and seen["Image_ori_array"][0].tolist() == [[1, 2, 3]]


def test_batched_rows_use_generate_batch():
    infer, pipe, _ = make()
    ids = torch.arange(12).view(2, 6) + 3
    infer.processor.prompt_len = 6
    res = infer(inputs={"input_ids": ids, "attention_mask": torch.ones_like(ids)})
    assert isinstance(res, list) and len(res) == 2 and len(pipe.calls) == 1 and len(pipe.calls[0][1]) == 2   # ONE pipeline call for both rows
    assert [r.predictions["IMAGE"] for r in res] == [[f"img:{c}"] for c in pipe.calls[0][1]]


def test_pipelining_state_machine():
    infer, pipe, thinker = make()
    infer.processor.prompt_len = 4
    reqs = [{"input_ids": torch.full((1, 4), 10 + i), "attention_mask": torch.ones(1, 4, dtype=torch.long)} for i in range(5)]
    serial = [infer.predict(inputs=r).response for r in reqs]
    infer2, pipe2, thinker2 = make()
    infer2.processor.prompt_len = 4
    outs = []
    for r in reqs:
        o = infer2.submit(inputs=r)
        outs.append(o)
    outs.append(infer2.flush())
    assert outs[0] is None and [o.response for o in outs[1:]] == serial      # request order, one step late
    assert len(thinker2.calls) == 5 and len(pipe2.calls) == 5                 # exactly one LLM pass and one decoder pass per request
    main = threading.current_thread().name
    # request 1's LLM pass and request 0's decoder pass: geometry not seen before -> both on the calling thread; from then on the
    # decoder pass has the helper thread
    assert [c[0] for c in thinker2.calls] == [main] * 5
    assert [c[0] for c in pipe2.calls] == [main, "spider-decoder-enqueue", "spider-decoder-enqueue", "spider-decoder-enqueue", main]
    assert infer2.flush() is None
    assert [r.response for r in infer2.pipelined(reqs)] == serial
    # a new LLM geometry (2 rows) drains through the non-overlapped path
    two = {"input_ids": torch.full((2, 4), 50), "attention_mask": torch.ones(2, 4, dtype=torch.long)}
    assert infer2.submit(inputs=reqs[0]) is None
    n = len(pipe2.calls)
    r0 = infer2.submit(inputs=two)
    assert r0.response == serial[0] and pipe2.calls[n][0] == main
    assert len(infer2.flush()) == 2
    # the decoders keep ONE captured geometry: after the 2-row decoder pass a 1-row one runs alone again, the one after it overlaps
    assert infer2.submit(inputs=reqs[0]) is None
    n = len(pipe2.calls)
    assert infer2.submit(inputs=reqs[1]).response == serial[0] and pipe2.calls[n][0] == main
    assert infer2.submit(inputs=reqs[2]).response == serial[1] and pipe2.calls[n + 1][0] == "spider-decoder-enqueue"
    assert infer2.flush().response == serial[2]


def test_errors():
    infer, _, _ = make()
    with pytest.raises(ValueError):
        infer()
    with pytest.raises(ValueError):
        SpiderFreeInfer(FakeThinker(), SyntheticOmniProcessor())
    infer.spider_decoder_infer.spider_decoder._pipes["IMAGE"] = lambda **kw: (_ for _ in ()).throw(RuntimeError("decoder failed"))
    infer.submit(inputs={"input_ids": torch.ones(1, 4, dtype=torch.long)})
    infer.submit(inputs={"input_ids": torch.ones(1, 4, dtype=torch.long)}) if False else None
    with pytest.raises(RuntimeError, match="decoder failed"):
        infer.flush()


def test_depth3_pipelining_state_machine():
    """depth 3: [decode loop of k+1 | decoder pass of k, then the prompt pass of k+2]; results in request order, two submits late;
    every request exactly one prompt pass, one decode loop, one decoder pass; KV cache sets alternate; cold steps on one thread."""
    pipe, thinker = FakePipe(), FakeThinker3()
    dinf = SpiderDecoderInfer({"model": dict(type="spider_decoder", pipelines={"IMAGE": pipe}, device="cpu")})
    proc = SyntheticOmniProcessor(vocab=500, head=3, prompt_len=4)
    infer = SpiderFreeInfer(thinker, proc, dinf, device="cpu", depth=3)
    ref, pipe_r, _ = make()
    ref.processor.prompt_len = 4
    reqs = [{"input_ids": torch.full((1, 4), 10 + i), "attention_mask": torch.ones(1, 4, dtype=torch.long)} for i in range(7)]
    serial = [ref.predict(inputs=r).response for r in reqs]
    outs = [infer.submit(inputs=r) for r in reqs]
    assert outs[0] is None and outs[1] is None and [o.response for o in outs[2:]] == serial[:5]
    with pytest.raises(RuntimeError, match="flush"):
        infer.predict(inputs=reqs[0])
    tail = []
    while True:
        r = infer.flush()
        if r is None:
            break
        tail.append(r.response)
    assert tail == serial[5:]
    main, helper = threading.current_thread().name, "spider-decoder-enqueue"
    assert [p[2] for p in thinker.pre] == [10 + i for i in range(7)] and [d[2] for d in thinker.dec] == [10 + i for i in range(7)]
    assert [p[1] for p in thinker.pre] == [1] * 7         # every prompt pass fills the staging KV cache set ...
    assert [d[1] for d in thinker.dec] == [0] * 7 and thinker.adopted >= 7      # ... and every decode loop runs from set 0 after an adopt
    # submits 1-3 are cold (prompt pass / + first decode graph of set 0 / + decoder graphs and decode graph of set 1): calling thread;
    # from the 4th on the decoder pass and the NEW request's prompt pass run on the helper thread, the decode loop on the calling one
    assert [p[0] for p in thinker.pre] == [main, main, main, helper, helper, helper, helper]
    assert [d[0] for d in thinker.dec] == [main] * 7
    assert [c[0] for c in pipe.calls] == [main, helper, helper, helper, helper, main, main]
    assert list(r.response for r in infer.pipelined(reqs[:4])) == serial[:4]
    with pytest.raises(ValueError):
        SpiderFreeInfer(FakeThinker(), proc, dinf, device="cpu", depth=3)


def test_pipelined_groups_conversations_into_batched_requests():
    """`pipelined(requests, group=n)`: consecutive conversations are answered together (one LLM pass with n left-padded rows, one
    decoder pass), every conversation still gets its own result in request order; a trailing partial group and processor-output
    dicts in between are handled; group=1 is the plain form."""
    infer, pipe, thinker = make()
    convs = [[{"role": "user", "content": " ".join(["word"] * (2 + 3 * i))}] for i in range(5)]       # different lengths: padding matters
    singles = [infer.predict(messages=m) for m in convs]
    infer2, pipe2, thinker2 = make()
    got = list(infer2.pipelined(convs, group=2))
    assert len(got) == 5 and all(not isinstance(r, list) for r in got)
    # rows are left-padded: each row's generated part starts after its assistant marker, so the per-conversation responses are the
    # ones the single calls produce (the fake thinker derives its tokens from the first column: rows of a batch share it with ... pad)
    assert [tuple(c[1]) for c in thinker2.calls] == [(2, thinker2.calls[0][1][1]), (2, thinker2.calls[1][1][1]), (1, thinker2.calls[2][1][1])]
    assert [len(c[1]) for c in pipe2.calls] == [2, 2, 1]                      # one decoder pass per group, one caption per row
    assert [r.predictions_text["IMAGE"] for r in got[4:]] == [singles[4].predictions_text["IMAGE"]]      # the un-padded last one is identical
    # left padding: the mask of the shorter row of a pair starts with zeros, the longer row has none
    inputs = infer2.build_inputs_batch(convs[:2])
    am = inputs["attention_mask"]
    assert am.shape[0] == 2 and int(am[1].min()) == 1 and int(am[0, 0]) == 0 and int(am[0, -1]) == 1
    assert int(inputs["input_ids"][0, -1]) == infer2.processor.ASSISTANT and int(inputs["input_ids"][1, -1]) == infer2.processor.ASSISTANT
    # dict requests pass through and flush the open group first; order is kept
    infer3, pipe3, _ = make()
    one = infer3.build_inputs(convs[0])
    mixed = [convs[0], one, convs[1], convs[2]]
    out3 = list(infer3.pipelined(mixed, group=2))
    assert len(out3) == 4 and [len(c[1]) for c in pipe3.calls] == [1, 1, 2]
    assert [r.response for r in infer3.pipelined(convs, group=1)] == [r.response for r in make()[0].pipelined(convs)]
    with pytest.raises(ValueError):
        list(infer3.pipelined(convs, group=0))


def test_build_inputs_batch_switches_a_right_padding_tokenizer_to_the_left():
    infer, _, _ = make()

    class Tok:
        padding_side = "right"
    infer.processor.tokenizer = Tok()
    infer.build_inputs_batch([[{"role": "user", "content": "a"}], [{"role": "user", "content": "b c d"}]])
    assert infer.processor.tokenizer.padding_side == "left"


# ------------------------------------------------------------------------------------------ round 5: contract gaps + advisor items
def test_story_mode_runs_story_generation_from_the_response(monkeypatch):
    """MODEL_NAME == "spider_story_free_qwen" (qwen2.5omni_spider_web.py:476-488): extract_story_elements -> story_generation"""
    import spider_amd.story as story
    calls = []
    monkeypatch.setattr(story, "story_generation", lambda pipe, general_prompt, prompt_array, style_name, **kw: calls.append(
        (pipe, general_prompt, prompt_array, style_name, kw)) or [f"panel:{p}" for p in prompt_array])

    class Proc(SyntheticOmniProcessor):
        def batch_decode(self, ids, **kw):
            return ["x\n<think>plan</think><GENERALPROMPT>a fox</GENERALPROMPT><PROMPTARRAY>['wakes up', 'eats']</PROMPTARRAY><STYLENAME>Comic book</STYLENAME>"]
    infer = SpiderFreeInfer(FakeThinker(), Proc(vocab=500), device="cpu", mode="spider_story_free_qwen", story_pipe="PIPE",
                            story_kwargs={"seed": 3})
    res = infer([{"role": "user", "content": "a day of a fox"}])
    assert calls == [("PIPE", "a fox", ["wakes up", "eats"], "Comic book", {"seed": 3})]
    assert res.predictions["IMAGESTORY"] == [["panel:wakes up", "panel:eats"]] and res.predictions_text["IMAGESTORY_prompts"] == [["wakes up", "eats"]]
    assert res.answers == [res.response]
    # elements missing: the reference's error line, no story
    infer.processor = SyntheticOmniProcessor(vocab=500, tags=("IMAGE",), head=3)
    res = infer([{"role": "user", "content": "x"}])
    assert res.predictions["IMAGESTORY"] == [] and len(calls) == 1
    with pytest.raises(ValueError):
        SpiderFreeInfer(FakeThinker(), Proc(vocab=500), device="cpu", mode="nope")


def test_mask_box_inputs_are_forwarded_and_rows_get_their_own_image():
    infer, pipe, _ = make(mask_box_inputs=lambda arr: {"IMAGE_SAM": [arr * 2], "Meta_info": {"original_shape": [arr.shape[:2]]}, "other": 1})
    infer.process_mm_info = lambda messages, use_audio_in_video: (None, [[[1, 2, 3]]], None)
    seen = {}
    orig = infer.spider_decoder_infer.spider_decoder.generate
    infer.spider_decoder_infer.spider_decoder.generate = lambda samples, *a: (seen.update(samples), orig(samples, *a))[1]
    infer([{"role": "user", "content": [{"type": "image", "image": "a.png"}, {"type": "text", "text": "x"}]}])
    assert seen["IMAGE_SAM"][0].tolist() == [[2, 4, 6]] and seen["Meta_info"] == {"original_shape": [(1, 3)]} and "other" not in seen
    # a batch of three conversations: 2 images, none, 1 image -> every row its OWN first image, none for the row without
    convs = [[{"role": "user", "content": [{"type": "image", "image": "a"}, {"type": "image", "image": "b"}]}],
             [{"role": "user", "content": "no picture"}],
             [{"role": "user", "content": [{"type": "image", "image": "c"}]}]]
    infer.process_mm_info = lambda conversations, use_audio_in_video: (None, ["A", "B", "C"], None)
    inputs = infer.build_inputs_batch(convs)
    assert inputs["_images"] == ["A", None, "C"]
    assert SpiderFreeInfer._n_images(convs[0]) == 2 and SpiderFreeInfer._n_images(convs[1]) == 0


def test_overlap_decision_comes_from_the_engines_not_from_shapes():
    """advisor (round 4): two audio clips of different lengths have the same tensor shapes; the thinker's graph caches are keyed by
    values. A thinker that reports would_capture() keeps such a pass off the overlapped schedule."""
    infer, pipe, thinker = make()
    infer.processor.prompt_len = 4
    cold = {"v": True}
    thinker.would_capture = lambda cache_set=0, decode=True, **kw: cold["v"]
    threads = []
    orig = thinker.generate
    thinker.generate = lambda *a, **k: (threads.append(threading.current_thread().name), orig(*a, **k))[1]
    dec_threads = []
    orig_dec = infer.decoder_pass
    infer.decoder_pass = lambda *a: (dec_threads.append(threading.current_thread().name), orig_dec(*a))[1]
    mk = lambda i: {"input_ids": torch.full((1, 4), 10 + i), "attention_mask": torch.ones(1, 4, dtype=torch.long)}
    for i in range(3):
        infer.submit(inputs=mk(i))
    assert all(t != "spider-decoder-enqueue" for t in dec_threads)          # cold every time: never overlapped, same shapes or not
    cold["v"] = False
    infer.submit(inputs=mk(3))
    assert dec_threads[-1] == "spider-decoder-enqueue"                        # the engine says its graphs exist: overlapped
    infer.flush()


def test_failed_decoder_pass_does_not_lose_the_next_request():
    infer, pipe, thinker = make()
    infer.processor.prompt_len = 4
    mk = lambda i: {"input_ids": torch.full((1, 4), 10 + i), "attention_mask": torch.ones(1, 4, dtype=torch.long)}
    for i in range(3):                                                          # warm: request 2 is pending, overlap is on
        infer.submit(inputs=mk(i))
    boom = {"on": True}
    orig = pipe.__class__.__call__

    def failing(self, prompt=None, **kw):
        if boom["on"]:
            boom["on"] = False
            raise RuntimeError("decoder blew up")
        return orig(self, prompt=prompt, **kw)
    pipe.__class__.__call__ = failing
    try:
        with pytest.raises(RuntimeError, match="decoder blew up"):
            infer.submit(inputs=mk(3))                                          # decoder pass of request 2 fails beside the LLM pass of 3
        nxt = infer.flush()                                                     # request 3's LLM result was kept: its decoder pass runs now
        assert nxt is not None and nxt.text_ids[0] == 13
    finally:
        pipe.__class__.__call__ = orig


def test_reset_warm_after_an_engine_swap_runs_the_next_passes_alone():
    """a host that swaps the engine behind a pipeline (bench.py's precise-mode leg) says so: the new engine has no graphs, its first
    decoder pass must not be enqueued beside another thread's LLM pass"""
    infer, pipe, thinker = make()
    infer.processor.prompt_len = 4
    reqs = [{"input_ids": torch.full((1, 4), 10 + i), "attention_mask": torch.ones(1, 4, dtype=torch.long)} for i in range(6)]
    main = threading.current_thread().name
    for r in reqs[:3]:
        infer.submit(inputs=r)
    assert pipe.calls[-1][0] == "spider-decoder-enqueue"                       # steady state: overlapped
    with pytest.raises(RuntimeError, match="in flight"):
        infer.reset_warm()
    assert infer.flush() is not None and infer.flush() is None
    infer.reset_warm()
    n = len(pipe.calls)
    assert infer.submit(inputs=reqs[3]) is None
    assert infer.submit(inputs=reqs[4]) is not None and pipe.calls[n][0] == main    # first decoder pass after the reset: calling thread
    assert infer.submit(inputs=reqs[5]) is not None and pipe.calls[n + 1][0] == "spider-decoder-enqueue"
    assert infer.flush() is not None


def _mk(i):
    return {"input_ids": torch.full((1, 4), 10 + i), "attention_mask": torch.ones(1, 4, dtype=torch.long)}


def test_cold_check_uses_the_generate_kwargs_the_real_pass_uses():
    """advisor (round 5): generate(**inputs, **generate_kwargs) picks the decode graph by output_hidden_states / return_logits; the
    coldness check must ask for THAT graph, and a key this caller has never run (or reset_warm() since) is cold whatever the engine says"""
    infer, pipe, thinker = make()
    infer.generate_kwargs = dict(output_hidden_states=True, spk="Chelsie")
    infer.processor.prompt_len = 4
    seen = []
    thinker.would_capture = lambda cache_set=0, decode=True, **kw: (seen.append(kw), False)[1]
    dec_threads = []
    orig_dec = infer.decoder_pass
    infer.decoder_pass = lambda *a: (dec_threads.append(threading.current_thread().name), orig_dec(*a))[1]
    infer.submit(inputs=_mk(0))
    infer.submit(inputs=_mk(1))               # the decoder geometry has not run yet: alone
    infer.submit(inputs=_mk(2))
    assert dec_threads[-1] == "spider-decoder-enqueue"
    assert seen and all(kw.get("output_hidden_states") is True and kw.get("spk") == "Chelsie" and "_images" not in kw for kw in seen)
    infer.flush()
    infer.reset_warm()                        # the engine still answers "warm": the caller's own record decides
    infer.submit(inputs=_mk(3))
    infer.submit(inputs=_mk(4))
    assert dec_threads[-1] != "spider-decoder-enqueue"
    infer.flush()


def test_failed_llm_pass_leaves_no_decoded_request_in_the_pipeline():
    """advisor (round 5): the LLM pass of request k+1 raises beside the decoder pass of request k -> k must not come back a second
    time, the helper's result rides on the exception, and the pipeline is empty afterwards"""
    infer, pipe, thinker = make()
    infer.processor.prompt_len = 4
    for i in range(3):
        infer.submit(inputs=_mk(i))
    orig = thinker.generate
    thinker.generate = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("llm blew up"))
    n = len(pipe.calls)
    with pytest.raises(RuntimeError, match="llm blew up") as ei:
        infer.submit(inputs=_mk(3))
    assert len(pipe.calls) == n + 1 and ei.value.spider_result.text_ids[0] == 12      # request 2 was decoded once, result not lost
    thinker.generate = orig
    assert infer.flush() is None and len(pipe.calls) == n + 1                           # ... and never again
    assert infer.submit(inputs=_mk(4)) is None                                          # the pipeline restarts empty


def test_depth3_failed_prompt_pass_retry_does_not_replay_the_adopted_request():
    pipe, thinker = FakePipe(), FakeThinker3()
    dinf = SpiderDecoderInfer({"model": dict(type="spider_decoder", pipelines={"IMAGE": pipe}, device="cpu")})
    proc = SyntheticOmniProcessor(vocab=500, tags=("IMAGE",), head=3)
    infer = SpiderFreeInfer(thinker, proc, dinf, device="cpu", depth=3)
    infer.processor.prompt_len = 4
    for i in range(5):
        infer.submit(inputs=_mk(i))
    assert pipe.calls[-1][0] == "spider-decoder-enqueue"                                # steady state
    orig = thinker.prefill_begin
    thinker.prefill_begin = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("prompt pass blew up"))
    n_dec = len(thinker.dec)
    with pytest.raises(RuntimeError, match="prompt pass blew up"):
        infer.submit(inputs=_mk(5))          # helper fails inside the new prompt pass; the retry on the calling thread fails too
    thinker.prefill_begin = orig
    assert infer._prefilled is None and len(thinker.dec) == n_dec + 1                   # the adopted request ran its decode loop once
    out = infer.flush()                      # its decoder pass
    assert out is not None and infer.flush() is None and len(thinker.dec) == n_dec + 1


def test_image_url_items_count_as_images_and_a_drift_is_refused():
    infer, pipe, _ = make()
    convs = [[{"role": "user", "content": [{"type": "image_url", "image_url": "http://x/a.png"}, {"type": "text", "text": "a"}]}],
             [{"role": "user", "content": [{"type": "text", "text": "b"}]}],
             [{"role": "user", "content": [{"image_url": "http://x/c.png"}, {"type": "image", "image": "c2"}]}]]
    assert [SpiderFreeInfer._n_images(c) for c in convs] == [1, 0, 2]
    infer.process_mm_info = lambda conversations, use_audio_in_video: (None, ["A", "C", "C2"], None)
    assert infer.build_inputs_batch(convs)["_images"] == ["A", None, "C"]
    infer.process_mm_info = lambda conversations, use_audio_in_video: (None, ["A", "C"], None)
    with pytest.raises(ValueError, match="drift"):
        infer.build_inputs_batch(convs)


@pytest.mark.parametrize("depth", [2, 3])
@pytest.mark.parametrize("seed", range(30))
def test_pipeline_state_machine_under_random_failures(depth, seed):
    """Property test of submit / flush with failures injected at random into the LLM pass, the prompt pass, the decode loop and the
    decoder pass: whatever is raised, (1) no request is ever decoded twice, (2) results (returned, or riding on an exception) come
    back in request order, (3) a request whose own passes all succeeded and that was not in flight beside a failure of its OWN pass
    comes back exactly once, (4) flush() drains the pipeline and leaves it empty."""
    import random
    rng = random.Random(1000 * depth + seed)
    pipe = FakePipe()
    thinker = FakeThinker3() if depth == 3 else FakeThinker()
    dinf = SpiderDecoderInfer({"model": dict(type="spider_decoder", pipelines={"IMAGE": pipe}, device="cpu")})
    proc = SyntheticOmniProcessor(vocab=500, tags=("IMAGE",), head=3)
    infer = SpiderFreeInfer(thinker, proc, dinf, device="cpu", depth=depth)
    infer.processor.prompt_len = 4
    fail = {"llm": set(), "pre": set(), "dec": set(), "img": set()}
    n_req = 40
    for i in range(n_req):
        for k in fail:
            if rng.random() < 0.08:
                fail[k].add(i)
    rid = lambda input_ids: int(input_ids[0, 0]) - 10
    pre_fired = set()

    orig_gen = FakeThinker.generate
    def gen(self, input_ids, attention_mask=None, **kw):
        if depth == 2 and rid(input_ids) in fail["llm"]:
            raise RuntimeError(f"llm {rid(input_ids)}")
        return orig_gen(self, input_ids, attention_mask, **kw)
    thinker.generate = gen.__get__(thinker)
    if depth == 3:
        orig_pre, orig_fin = thinker.prefill_begin, thinker.decode_finish
        def pre(input_ids, attention_mask=None, cache_set=0, **kw):
            if rid(input_ids) in fail["pre"]:
                fail["pre"].discard(rid(input_ids))          # (fails once: the overlapped step's retry on the calling thread succeeds)
                pre_fired.add(rid(input_ids))
                raise RuntimeError(f"pre {rid(input_ids)}")
            return orig_pre(input_ids, attention_mask, cache_set=cache_set, **kw)
        def fin(handle):
            if rid(handle[1]) in fail["dec"]:
                raise RuntimeError(f"dec {rid(handle[1])}")
            return FakeThinker.generate(thinker, handle[1])
        thinker.prefill_begin, thinker.decode_finish = pre, fin
    decoded = []
    orig_call = FakePipe.__call__
    def img(self, prompt=None, **kw):
        r = None
        return orig_call(self, prompt=prompt, **kw)
    orig_dec = infer.decoder_pass
    def dec(text_ids, responses, images=None):
        r = int(text_ids[0][0]) - 10
        decoded.append(r)
        if r in fail["img"]:
            raise RuntimeError(f"img {r}")
        return orig_dec(text_ids, responses, images)
    infer.decoder_pass = dec
    got = []
    def take(res):
        if res is not None:
            got.append(int(res.text_ids[0]) - 10)
    for i in range(n_req):
        try:
            take(infer.submit(inputs=_mk(i)))
        except RuntimeError as e:
            take(getattr(e, "spider_result", None))
    for _ in range(8):
        try:
            r = infer.flush()
        except RuntimeError as e:
            take(getattr(e, "spider_result", None))
            continue
        if r is None:
            break
        take(r)
    assert infer.flush() is None and infer._pending is None and infer._prefilled is None
    assert len(decoded) == len(set(decoded)), f"a request was decoded twice: {decoded}"
    assert got == sorted(got) and len(got) == len(set(got)), f"results out of order or duplicated: {got}"
    bad = fail["llm"] | fail["dec"] | fail["img"] if depth == 2 else fail["dec"] | fail["img"]
    clean = [i for i in range(n_req) if i not in bad]
    lost = [i for i in clean if i not in got]
    # a clean request may only be lost when its own prompt pass failed on the one-thread path (no retry there) or when a NEIGHBOUR's
    # failure hit the step that carried it (its result rides on the exception when it was computed): nothing else may vanish
    near = fail["llm"] | fail["dec"] | fail["img"] | pre_fired
    assert all(i in pre_fired or any(abs(i - b) <= 2 for b in near) for i in lost), (lost, fail, pre_fired)


@pytest.mark.parametrize("depth", [2, 3])
@pytest.mark.parametrize("seed", range(10))
def test_pipelined_generator_random_request_mixes(depth, seed):
    """`pipelined(requests, group=n)` over random mixes of conversations (merged into batched requests of up to n rows) and
    processor-output dicts (passed through; several rows come back as a list): one result per conversation / dict, in request order,
    every request's LLM work done exactly once, the same responses as the one-request-at-a-time calls."""
    import random
    rng = random.Random(77 * depth + seed)
    pipe, thinker = FakePipe(), (FakeThinker3() if depth == 3 else FakeThinker())
    dinf = SpiderDecoderInfer({"model": dict(type="spider_decoder", pipelines={"IMAGE": pipe}, device="cpu")})
    proc = SyntheticOmniProcessor(vocab=500, tags=("IMAGE",), head=3)
    infer = SpiderFreeInfer(thinker, proc, dinf, device="cpu", depth=depth)
    ref = SpiderFreeInfer(FakeThinker(), SyntheticOmniProcessor(vocab=500, tags=("IMAGE",), head=3),
                          SpiderDecoderInfer({"model": dict(type="spider_decoder", pipelines={"IMAGE": FakePipe()}, device="cpu")}), device="cpu")
    reqs, want = [], []
    for i in range(rng.randint(1, 14)):
        conv = [{"role": "user", "content": " ".join([f"w{i}"] * rng.randint(1, 6))}]
        if rng.random() < 0.3:
            rows = rng.randint(1, 3)
            d = ref.build_inputs_batch([conv] * rows) if rows > 1 else ref.build_inputs(conv)
            reqs.append(dict(d))
            r = ref.predict(inputs=dict(d))
            want.append([x.response for x in r] if isinstance(r, list) else r.response)
        else:
            reqs.append(conv)
            want.append(None)                  # (a merged row's tokens depend on its batch's padding: checked by count and order only)
    group = rng.randint(1, 4)
    got = list(infer.pipelined(reqs, group=group))
    assert len(got) == len(reqs)
    for rq, w, g in zip(reqs, want, got):
        if isinstance(rq, dict):
            B = rq["input_ids"].shape[0]
            assert (isinstance(g, list) and len(g) == B) if B > 1 else not isinstance(g, list)
            assert ([x.response for x in g] if isinstance(g, list) else g.response) == w
        else:
            assert not isinstance(g, list) and g.answers == [g.response] and len(g.predictions["IMAGE"]) == 1
    n_rows = sum(rq["input_ids"].shape[0] if isinstance(rq, dict) else 1 for rq in reqs)
    rows_llm = sum(c[1][0] for c in thinker.calls)
    assert rows_llm == n_rows and sum(len(c[1]) for c in pipe.calls) == n_rows          # every row: one LLM pass, one caption decoded
    assert infer.flush() is None
