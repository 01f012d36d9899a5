"""CPU: the SpiderDecoder / SpiderDecoderInfer drop-in (registry construction, generate contract, fail-soft decoders,
story hand-off) with injected stand-in pipelines -- no GPU, no kernels."""
from spider_amd import routing
from spider_amd.registry import registry
from spider_amd.spider_decoder import SpiderDecoder, SpiderDecoderInfer  # noqa: F401  (registers "spider_decoder")


class FakePipe:
    def __init__(self, field):
        self.field, self.calls = field, []

    def __call__(self, prompt=None, prompt_embeds=None, return_prompts_only=False, **kw):
        self.calls.append((prompt, kw))
        class O:
            pass
        o = O()
        setattr(o, self.field, [f"{self.field}:{prompt[0]}"])
        return o


def test_registry_build_and_known_answer():
    cfg = dict(model=dict(type="spider_decoder", name="spider_decoder", diffusion_modules={}, mask_decoder_modules=None,
                          story_generation=None, max_context_len=4096))
    infer = SpiderDecoderInfer(cfg)
    assert isinstance(infer.spider_decoder, registry.get_model_class("spider_decoder"))
    # spider_decoder_infer.py:139-142 -- without checkpoints every decoder fails soft, the text routing still happens
    a, p, pt = infer({"llm_text_all": ["<IMAGE>apple</IMAGE><VIDEO>dog</VIDEO><AUDIO>cat</AUDIO>"]})
    assert a == ["<IMAGE>apple</IMAGE><VIDEO>dog</VIDEO><AUDIO>cat</AUDIO>"]
    assert pt == {'IMAGE': ['apple'], 'VIDEO': ['dog'], 'AUDIO': ['cat'], 'MASK': [], 'BOX': [], 'IMAGESTORY': [],
                  'IMAGESTORY_prompts': []}
    assert p["IMAGE"] == [] and p["VIDEO"] == [] and p["AUDIO"] == []


def test_dispatch_with_injected_pipelines_and_defaults():
    img, vid, aud = FakePipe("images"), FakePipe("frames"), FakePipe("audios")
    dec = SpiderDecoder(diffusion_modules={}, pipelines=dict(IMAGE=img, VIDEO=vid, AUDIO=aud))
    answers, predictions, ptext = routing.new_outputs()
    dec.generate({"llm_text_all": ["<AUDIO>rain</AUDIO><IMAGE>sun</IMAGE><IMAGE>moon</IMAGE><VIDEO>sea</VIDEO>"]},
                 answers, predictions, ptext)
    assert predictions["IMAGE"] == ["images:sun", "images:moon"]      # preds[0] per caption
    assert predictions["VIDEO"] == [["frames:sea"]]                    # whole frame list per caption
    assert predictions["AUDIO"] == ["audios:rain"]
    # reference defaults of the decode_* signatures (spider_decoder.py:100,122,145)
    assert img.calls[0][1] == dict(guidance_scale=7.5, num_inference_steps=40)
    assert vid.calls[0][1] == dict(guidance_scale=7.5, num_inference_steps=40, height=320, width=576, num_frames=16)
    assert aud.calls[0][1] == dict(guidance_scale=7.5, num_inference_steps=40, audio_length_in_s=5.0)


def test_story_handoff(monkeypatch):
    import spider_amd.story as story
    seen = {}
    def fake_story(pipe, general_prompt=None, prompt_array=None, style_name=None):
        seen.update(pipe=pipe, g=general_prompt, a=prompt_array, s=style_name)
        return ["img0", "img1"]
    monkeypatch.setattr(story, "story_generation", fake_story)
    infer = SpiderDecoderInfer(dict(model=dict(type="spider_decoder", diffusion_modules={})), story_pipe="PIPE")
    text = ("<IMAGESTORY><GENERALPROMPT> 'a man with a black suit' </GENERALPROMPT> <PROMPTARRAY> ['wake up in the bed', "
            "'have breakfast'] </PROMPTARRAY> <STYLENAME> 'Comic book' </STYLENAME></IMAGESTORY>")
    a, p, pt = infer({"llm_text_all": [text]})
    assert seen == dict(pipe="PIPE", g="'a man with a black suit'", a=["wake up in the bed", "have breakfast"], s="'Comic book'")
    assert p["IMAGESTORY"] == [["img0", "img1"]] and pt["IMAGESTORY_prompts"] == [["wake up in the bed", "have breakfast"]]


class FakeBatchPipe:
    """Stand-in pipeline that accepts a list of prompts (one pipeline call for the whole batch)."""
    def __init__(self, field):
        self.field, self.calls = field, []

    def __call__(self, prompt=None, prompt_embeds=None, return_prompts_only=False, **kw):
        import numpy as np
        self.calls.append((list(prompt), kw))
        class O:
            pass
        o = O()
        if self.field == "frames":   # tensor2vid layout: F frames of [H, B*W, 3], the batch tiled along the width
            n = len(prompt)
            setattr(o, self.field, [np.concatenate([np.full((2, 3, 3), 10 * j + f, dtype=np.uint8) for j in range(n)], 1) for f in range(4)])
        else:
            setattr(o, self.field, [f"{self.field}:{p}" for p in prompt])
        return o


def test_generate_batch_equals_per_sample_generate(golden_dir):
    """SpiderDecoder.generate_batch (the batched entry point): per-sample containers identical to one
    `generate` call per sample -- on the reference-generated routing cases -- while every diffusion modality runs ONE pipeline
    call over all captions of the batch, and the tiled video frames are cut back per caption."""
    import json, os
    cases = json.load(open(os.path.join(golden_dir, "routing_ref.json")))["cases"]
    texts = [c["text"] for c in cases]
    mk = lambda cls: dict(IMAGE=cls("images"), VIDEO=cls("frames"), AUDIO=cls("audios"))
    single_pipes, batch_pipes = mk(FakePipe), mk(FakeBatchPipe)
    d1 = SpiderDecoder(diffusion_modules={}, pipelines=single_pipes)
    d2 = SpiderDecoder(diffusion_modules={}, pipelines=batch_pipes)
    ref = [d1.generate({"llm_text_all": [t]}, *routing.new_outputs()) for t in texts]
    got = d2.generate_batch([{"llm_text_all": [t]} for t in texts])
    assert len(got) == len(ref)
    for (a1, p1, t1), (a2, p2, t2), c in zip(ref, got, cases):
        assert a1 == a2 and t1 == t2                                   # answers and predictions_text: identical
        assert t2 == c["predictions_text"] and a2 == c["answers"]      # ... and equal to what the reference's generate produced
        assert p1["IMAGE"] == p2["IMAGE"] and p1["AUDIO"] == p2["AUDIO"]
        assert len(p1["VIDEO"]) == len(p2["VIDEO"])
        for v in p2["VIDEO"]:
            assert len(v) == 4 and v[0].shape == (2, 3, 3)             # one caption's own frames, not the tiled batch
    n_img = sum(len(c["predictions_text"]["IMAGE"]) for c in cases)
    assert len(batch_pipes["IMAGE"].calls) == 1 and len(batch_pipes["IMAGE"].calls[0][0]) == n_img
    assert len(batch_pipes["AUDIO"].calls) == 1
    # videos go through the pipeline in chunks of `video_batch` captions (bounded activation tensors); the j-th caption of a call
    # gets the j-th width slice of every frame
    vids = [v for _, p, _ in got for v in p["VIDEO"]]
    assert len(batch_pipes["VIDEO"].calls) == (len(vids) + d2.video_batch - 1) // d2.video_batch
    assert [len(c[0]) for c in batch_pipes["VIDEO"].calls] == [min(d2.video_batch, len(vids) - i) for i in range(0, len(vids), d2.video_batch)]
    assert [int(v[0][0, 0, 0]) for v in vids] == [10 * (j % d2.video_batch) for j in range(len(vids))]
