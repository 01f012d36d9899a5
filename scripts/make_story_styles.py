"""Regenerates spider_amd/data/story_styles.json (prompt DATA: style name -> [positive template, negative prompt]) and the
fixture tests/golden/story_styles_ref.json (what Comic_Generation.py:408-413's apply_style / apply_style_positive return
for the canonical config-3 call) from the reference's StoryDiffusion/utils/style_template.py. Run in the build container only
(the reference tree does not travel)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("style_template", "/root/reference/StoryDiffusion/utils/style_template.py")
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
table = {k: [p, n] for k, (p, n) in m.styles.items()}
with open(os.path.join(ROOT, "spider_amd", "data", "story_styles.json"), "w") as f:
    json.dump(table, f, indent=1, ensure_ascii=False)

# expected strings of the reference's two closures (restated call, reference table) for every style + an unknown name
NEG = ("naked, deformed, bad anatomy, disfigured, poorly drawn face, mutation, extra limb, ugly, disgusting, poorly drawn hands, "
       "missing limb, floating limbs, disconnected limbs, blurry, watermarks, oversaturated, distorted hands, amputation")
positives = ["a man with a black suit,wake up in the bed", "a man with a black suit,have breakfast"]
cases = []
for name in list(m.styles) + ["no such style"]:
    p, n = m.styles.get(name, m.styles["(No style)"])
    cases.append(dict(style=name, positives=positives, negative=NEG, out_prompts=[p.replace("{prompt}", x) for x in positives],
                      out_negative=n + " " + NEG, out_single=p.replace("{prompt}", positives[1])))
with open(os.path.join(ROOT, "tests", "golden", "story_styles_ref.json"), "w") as f:
    json.dump(cases, f, indent=1, ensure_ascii=False)
print(len(table), "styles")
