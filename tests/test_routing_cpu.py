"""CPU: product-side routing (spider_amd/routing.py) and registry against the reference-generated golden file and
the reference's in-file known answers -- string/int exact."""
import json
import os

from spider_amd import routing
from spider_amd.registry import registry


def test_routing_matches_reference_golden(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "routing_ref.json")))
    for c in ref["cases"]:
        calls = []
        def fake(mod, ret_none=False):
            def f(samples, **kw):
                calls.append([mod, samples["llm_text_res"][0]])
                return None if ret_none else [f"{mod}:{samples['llm_text_res'][0]}"]
            return f
        def fake_box(samples):
            calls.append(["BOX", samples["llm_text_res"][0]])
            return dict(outputs_bboxes=[["bb"]], outputs_label_names=[["ln"]], outputs_scores=[[0.9]])
        nm = c["none_mode"]
        dm = dict(IMAGE=fake("IMAGE", nm), VIDEO=fake("VIDEO"), AUDIO=fake("AUDIO", nm), MASK=fake("MASK"), BOX=fake_box,
                  IMAGESTORY=None)
        answers, predictions, ptext = routing.new_outputs()
        a, p, pt = routing.route({"llm_text_all": [c["text"]]}, answers, predictions, ptext, dm)
        assert a is answers and p is predictions and pt is ptext      # caller-owned containers are mutated and returned
        assert a == c["answers"] and pt == c["predictions_text"] and p == c["predictions"] and calls == c["calls"], c["text"]
    for s in ref["story"]:
        assert routing.extract_story_elements(s["text"]) == (s["general_prompt"], s["prompt_array"], s["style_name"])
        assert routing.extract_answer(s["text"]) == s["answer"]


def test_routing_matches_reference_on_random_grammar_strings(golden_dir):
    """400 routing strings + 250 story strings drawn from a grammar (random tag order, nesting and damage, regex metacharacters,
    unicode, <think> blocks, every clean_prompt_array fallback), answered by the reference's own functions
    (tests/golden/make_golden.py routing_fuzz): containers, decoder-call order, story triples and extract_answer all exact."""
    ref = json.load(open(os.path.join(golden_dir, "routing_ref_fuzz.json")))
    assert len(ref["cases"]) == 400 and len(ref["story"]) == 250
    n_calls = 0
    for c in ref["cases"]:
        calls = []
        def fake(mod, ret_none=False):
            def f(samples, **kw):
                calls.append([mod, samples["llm_text_res"][0]])
                return None if ret_none else [f"{mod}:{samples['llm_text_res'][0]}"]
            return f
        def fake_box(samples):
            calls.append(["BOX", samples["llm_text_res"][0]])
            return dict(outputs_bboxes=[["bb"]], outputs_label_names=[["ln"]], outputs_scores=[[0.9]])
        nm = c["none_mode"]
        dm = dict(IMAGE=fake("IMAGE", nm), VIDEO=fake("VIDEO"), AUDIO=fake("AUDIO", nm), MASK=fake("MASK"), BOX=fake_box, IMAGESTORY=None)
        answers, predictions, ptext = routing.new_outputs()
        a, p, pt = routing.route({"llm_text_all": [c["text"]]}, answers, predictions, ptext, dm)
        assert a == c["answers"] and pt == c["predictions_text"] and p == c["predictions"] and calls == c["calls"], repr(c["text"])
        assert routing.get_llm_text_modality(c["text"], ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX"]) == c["modality"], repr(c["text"])
        for m, want in c["res"].items():
            assert routing.get_llm_text_res(c["text"], m) == want, (m, repr(c["text"]))
        n_calls += len(calls)
    assert n_calls > 300                                # the strings do exercise the decoders
    for s in ref["story"]:
        assert s["raises"] is None                      # (the reference never raised on these; a raise would be part of the contract)
        assert routing.extract_story_elements(s["text"]) == (s["general_prompt"], s["prompt_array"], s["style_name"]), repr(s["text"])
        assert routing.extract_answer(s["text"]) == s["answer"], repr(s["text"])


def test_known_answers_and_registry():
    a, pt, calls = routing.route_text("<IMAGE>apple</IMAGE><VIDEO>dog</VIDEO><AUDIO>cat</AUDIO>")
    assert pt == {'IMAGE': ['apple'], 'VIDEO': ['dog'], 'AUDIO': ['cat'], 'MASK': [], 'BOX': [], 'IMAGESTORY': [],
                  'IMAGESTORY_prompts': []}
    assert calls == [("IMAGE", "apple"), ("VIDEO", "dog"), ("AUDIO", "cat")]
    # dispatch follows dict-key order, not text order
    _, _, calls = routing.route_text("<AUDIO>rain</AUDIO> first then <IMAGE>sun</IMAGE>")
    assert calls == [("IMAGE", "sun"), ("AUDIO", "rain")]

    @registry.register_model("unit_test_model")
    class M:
        def __init__(self, a=1):
            self.a = a
    assert registry.get_model_class("unit_test_model")(a=3).a == 3
    assert registry.get_model_class("missing") is None
