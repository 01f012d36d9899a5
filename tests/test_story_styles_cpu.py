"""CPU: the StoryDiffusion style table shipped with the package reproduces, string for string, what the reference's
apply_style / apply_style_positive closures (Comic_Generation.py:408-413) return with its utils/style_template.py
(fixture: tests/golden/story_styles_ref.json, written by scripts/make_story_styles.py from the reference's table)."""
import json
import os
import warnings

import pytest

from spider_amd import story


def test_styles_reproduce_reference_strings(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "story_styles_ref.json")))
    assert len(cases) == 10 and any(c["style"] == "Comic book" for c in cases)
    for c in cases:
        known = c["style"] in story.styles()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            prompts, neg = story.apply_style(c["style"], c["positives"], c["negative"])
            single = story.apply_style_positive(c["style"], c["positives"][1])
        assert prompts == c["out_prompts"] and neg == c["out_negative"] and single == c["out_single"], c["style"]
        assert bool(w) == (not known)          # an unknown style falls back like the reference, but says so
    # the canonical config-3 style (train_configs/spider_story_free_llama3.py:11) is not the empty template
    p, n = story.apply_style("Comic book", ["x"], "y")
    assert p == ["comic x . graphic illustration, comic art, graphic novel art, vibrant, highly detailed"] and n.endswith(" y") and len(n) > 50
    assert story.NEGATIVE_PROMPT == cases[0]["negative"]


def test_unknown_style_warns():
    with pytest.warns(UserWarning, match="not in the style table"):
        assert story.apply_style_positive("nope", "a cat") == "a cat"
