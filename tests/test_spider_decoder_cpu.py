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
