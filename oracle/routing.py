"""ORACLE (test infrastructure, not product code): restatement of the reference's signal-tag routing.

Follows
  get_llm_text_res        spider/models/spider_decoder.py:283-291
  get_llm_text_modality   spider/models/spider_decoder.py:293-306
  generate (routing part) spider/models/spider_decoder.py:309-348  (dict-key order IMAGE,VIDEO,AUDIO,MASK,BOX,
                          IMAGESTORY; caption appended to predictions_text before the decoder runs; a decoder
                          returning None is skipped)
  clean_prompt_array      spider_decoder_infer.py:86-112
  extract_story_elements  spider_decoder_infer.py:114-129 (= demo/inference_api.py:178-221)
  extract_answer          qwen2.5omni_spider_web.py:341-347

Pinned against tests/golden/routing_ref.json (produced by tests/golden/make_golden.py from the reference's
own functions) and the in-file known answers spider_decoder_infer.py:139-142, spider_decoder.py:284-295.
"""
import ast
import json
import re

MODALITY_KEYS = ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX", "IMAGESTORY"]


def get_llm_text_res(string, modality):
    return re.findall(rf"<{modality}>(.*?)</{modality}>", string)


def get_llm_text_modality(string, modality_keys):
    return [m for m in modality_keys if re.search(rf"<{m}>.*?</{m}>", string)]


def route(text, decoders=None):
    """Returns (answers, predictions_text, call_order) for one LLM text; decoders: name -> callable(caption)
    returning a prediction or None (None is skipped)."""
    predictions_text = {k: [] for k in MODALITY_KEYS + ["IMAGESTORY_prompts"]}
    predictions = {k: [] for k in MODALITY_KEYS}
    calls = []
    for m in get_llm_text_modality(text, MODALITY_KEYS):
        for cap in get_llm_text_res(text, m):
            predictions_text[m].append(cap)
            if m == "IMAGESTORY":
                continue
            calls.append((m, cap))
            if decoders and m in decoders:
                r = decoders[m](cap)
                if r is not None:
                    predictions[m].append(r)
    return [text], predictions_text, calls, predictions


def clean_prompt_array(prompt_str):
    if not prompt_str.strip():
        return []
    prompt_str = re.sub(r"<.*?>", "", prompt_str).strip()
    try:
        parsed = ast.literal_eval(prompt_str)
        if isinstance(parsed, list):
            return [str(i).strip() for i in parsed if i]
    except (SyntaxError, ValueError):
        pass
    try:
        parsed = json.loads(prompt_str)
        if isinstance(parsed, list):
            return [str(i).strip() for i in parsed if i]
    except json.JSONDecodeError:
        pass
    prompt_str = re.sub(r"^\[|\]$", "", prompt_str.strip())
    prompts = re.split(r"'\s*,\s*'|\"\s*,\s*\"|\n", prompt_str)
    return [p.strip(" '\"") for p in prompts if p.strip()]


def extract_story_elements(output_texts):
    sp = output_texts.split("</think>", 1)
    if len(sp) > 1:
        output_texts = sp[1]
    g = re.findall(r"<GENERALPROMPT>\s*(.*?)\s*</GENERALPROMPT>", output_texts, re.DOTALL)
    general = g[-1].strip() if g else ""
    a = re.findall(r"<PROMPTARRAY>\s*(.*?)\s*</PROMPTARRAY>", output_texts, re.DOTALL)
    arr = clean_prompt_array(a[-1].strip() if a else "[]")
    s = re.findall(r"<STYLENAME>\s*(.*?)\s*</STYLENAME>", output_texts, re.DOTALL)
    style = s[-1].strip() if s else ""
    return general, arr, style
