"""Signal-tag routing of the LLM text to the modality decoders (pure host string work, exact).

Mirrors the reference's Decoders-Controller:
    get_llm_text_res / get_llm_text_modality   spider/models/spider_decoder.py:283-306
    the routing loop of SpiderDecoder.generate spider/models/spider_decoder.py:309-348
    clean_prompt_array / extract_story_elements spider_decoder_infer.py:86-129 (= demo/inference_api.py:178-221)
    extract_answer                              qwen2.5omni_spider_web.py:341-347
Semantics that must be kept bit-exact: non-greedy `<M>(.*?)</M>` without DOTALL; modalities visited in dict-key
order IMAGE, VIDEO, AUDIO, MASK, BOX, IMAGESTORY (not text order); caption appended to predictions_text before the
decoder runs; a decoder returning None is skipped; story triple = LAST match after the first </think>.
"""
import ast
import json
import re
from typing import Callable, Dict, List, Optional, Tuple

MODALITY_KEYS = ["IMAGE", "VIDEO", "AUDIO", "MASK", "BOX", "IMAGESTORY"]


def get_llm_text_res(string: str, modality: str) -> List[str]:
    return re.findall(rf"<{modality}>(.*?)</{modality}>", string)


def get_llm_text_modality(string: str, modality_keys) -> List[str]:
    return [m for m in modality_keys if re.search(rf"<{m}>.*?</{m}>", string)]


def new_outputs() -> Tuple[list, dict, dict]:
    """The three caller-owned containers of the generate contract (spider_decoder_infer.py:49-66)."""
    answers: list = []
    predictions = dict(IMAGE=[], VIDEO=[], AUDIO=[], MASK=[], BOX=dict(bboxes=[], label_names=[], scores=[]), IMAGESTORY=[])
    predictions_text = dict(IMAGE=[], VIDEO=[], AUDIO=[], MASK=[], BOX=[], IMAGESTORY=[], IMAGESTORY_prompts=[])
    return answers, predictions, predictions_text


def route(samples: dict, answers: list, predictions: dict, predictions_text: dict,
          decode_modality: Dict[str, Optional[Callable]]):
    """The body of SpiderDecoder.generate (spider_decoder.py:310-348): mutates and returns the caller's containers."""
    output_texts = samples["llm_text_all"][0]
    for modality in get_llm_text_modality(output_texts, decode_modality.keys()):
        for llm_text_res in get_llm_text_res(output_texts, modality):
            samples["llm_text_res"] = [llm_text_res]
            predictions_text[modality].append(llm_text_res)
            if modality in ("IMAGE", "AUDIO", "MASK"):
                preds = decode_modality[modality](samples)
                if preds is not None:
                    predictions[modality].append(preds[0])
            elif modality == "VIDEO":
                preds = decode_modality[modality](samples)
                if preds is not None:
                    predictions[modality].append(preds)
            elif modality == "BOX":
                det = decode_modality[modality](samples)
                if det is not None:
                    predictions[modality]["bboxes"].append(det["outputs_bboxes"][0])
                    predictions[modality]["label_names"].append(det["outputs_label_names"][0])
                    predictions[modality]["scores"].append(det["outputs_scores"][0])
    answers.append(output_texts)
    return answers, predictions, predictions_text


def route_batch(samples_list: List[dict], outputs: List[tuple], decode_modality: Dict[str, Optional[Callable]],
                batch_decoders: Dict[str, Callable]):
    """The routing loop of SpiderDecoder.generate for SEVERAL independent samples at once (the batched
    entry point offered next to the reference's one-sample contract). Result = calling `route` once per sample with that sample's
    own containers -- same captions in the same order, same None-skip rule -- except that every caption of a diffusion modality
    (IMAGE / VIDEO / AUDIO), over all samples, goes to ONE call of `batch_decoders[modality](captions) -> list | None` (one
    entry per caption, shaped like the single call's `preds[0]` (IMAGE, AUDIO) or `preds` (VIDEO)). MASK / BOX keep their
    per-sample calls (they read the sample's own image)."""
    jobs: Dict[str, list] = {}
    for i, (samples, (answers, predictions, predictions_text)) in enumerate(zip(samples_list, outputs)):
        output_texts = samples["llm_text_all"][0]
        for modality in get_llm_text_modality(output_texts, decode_modality.keys()):
            for llm_text_res in get_llm_text_res(output_texts, modality):
                predictions_text[modality].append(llm_text_res)
                if modality in batch_decoders:
                    jobs.setdefault(modality, []).append((i, llm_text_res))
                    continue
                samples["llm_text_res"] = [llm_text_res]
                if modality == "MASK":
                    preds = decode_modality[modality](samples)
                    if preds is not None:
                        predictions[modality].append(preds[0])
                elif modality == "BOX":
                    det = decode_modality[modality](samples)
                    if det is not None:
                        predictions[modality]["bboxes"].append(det["outputs_bboxes"][0])
                        predictions[modality]["label_names"].append(det["outputs_label_names"][0])
                        predictions[modality]["scores"].append(det["outputs_scores"][0])
        answers.append(output_texts)
    for modality in [m for m in decode_modality.keys() if m in jobs]:       # dict-key order, as the single-sample loop visits them
        res = batch_decoders[modality]([c for _, c in jobs[modality]])
        if res is None:
            continue
        assert len(res) == len(jobs[modality]), f"{modality}: the batched decoder must return one entry per caption"
        for (i, caption), r in zip(jobs[modality], res):
            outputs[i][1][modality].append(r)
            samples_list[i]["llm_text_res"] = [caption]
    return outputs


def route_text(text: str):
    """Routing without decoders: (answers, predictions_text, [(modality, caption), ...] in dispatch order)."""
    calls = []
    def rec(m):
        def f(samples):
            calls.append((m, samples["llm_text_res"][0]))
            return None
        return f
    answers, predictions, ptext = new_outputs()
    dm = {m: (rec(m) if m != "IMAGESTORY" else None) for m in MODALITY_KEYS}
    route({"llm_text_all": [text]}, answers, predictions, ptext, dm)
    return answers, ptext, calls


def extract_answer(output_texts: str) -> str:
    sp = output_texts.split("</think>", 1)
    return sp[1] if len(sp) > 1 else output_texts


def clean_prompt_array(prompt_str: str) -> List[str]:
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


def extract_story_elements(output_texts: str):
    output_texts = extract_answer(output_texts)
    g = re.findall(r"<GENERALPROMPT>\s*(.*?)\s*</GENERALPROMPT>", output_texts, re.DOTALL)
    a = re.findall(r"<PROMPTARRAY>\s*(.*?)\s*</PROMPTARRAY>", output_texts, re.DOTALL)
    s = re.findall(r"<STYLENAME>\s*(.*?)\s*</STYLENAME>", output_texts, re.DOTALL)
    return (g[-1].strip() if g else "", clean_prompt_array(a[-1].strip() if a else "[]"), s[-1].strip() if s else "")
