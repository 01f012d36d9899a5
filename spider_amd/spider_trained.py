"""Trained-Spider `generate` on the native engines, keeping the reference's contract

    Spider.generate(samples, answers, predictions, predictions_text) -> (answers, predictions, predictions_text)

(spider/models/spider.py:1465-1621): prompt embedding (`prepare_generation_embedding` :1623-1673, text-only questions),
greedy LLM decode with hidden-state capture, signal-token routing of the generated text, capture of the LLM states at
the `[M0]` signal tokens (`preparing_output_embeds_infer` :1413-1463), projection through TextFcLayerMoE
(spider_amd/moe_proj.py), the 0.1 / 0.9 blend with the diffusion text encoder's embedding (:420,432,444) and the
`prompt_embeds` entry of the diffusion pipelines (decode_image / decode_video / decode_audio :346-520).

Differences, all deliberate: pipelines are built once and reused (the reference reloads the checkpoint in every
decode call); multimodal INPUT placeholders (`<IMAGE-Placeholder>` -> ImageBind encoders, :1642-1649) are outside this
path and raise NotImplementedError; MASK / BOX decoders fail soft as in spider_amd/spider_decoder.py."""
from __future__ import annotations

import re
from typing import Dict, List, Optional

import torch

from . import ops, routing
from .llm import StoppingCriteriaSub

BF16 = torch.bfloat16


class TrainedSpider:
    def __init__(self, llm, tokenizer, alignment_projs: List, output_alignment_modules: Dict[str, dict],
                 modality_tokens: Dict[str, int], pipelines: Optional[Dict[str, object]] = None, max_context_len: int = 100,
                 output_alignment_MoE_mode: Optional[str] = "moe_transformer", using_lora: bool = False):
        """llm: spider_amd.llm.LlamaEngine; tokenizer: the LLM tokenizer with the signal tokens added;
        alignment_projs: one TextFcLayerMoE per entry of output_alignment_modules[M]['alignment_layer'] (shared by all
        modalities, as in the reference's MoE mode), or -- output_alignment_MoE_mode=None -- {modality: [TextFcLayer, ...]}; modality_tokens: tokenizer_modules['new_modality_tokens']
        (spider.py:142); pipelines: {'IMAGE'|'VIDEO'|'AUDIO': pipeline object}."""
        if output_alignment_MoE_mode is None and not isinstance(alignment_projs, dict):
            raise ValueError("output_alignment_MoE_mode=None: alignment_projs must be {modality: [TextFcLayer per alignment layer]} "
                             "(spider.py:200-209)")
        self.output_alignment_MoE_mode = output_alignment_MoE_mode
        self.llama_model, self.llama_tokenizer = llm, tokenizer
        self.alignment_projs = alignment_projs
        self.output_alignment_modules = output_alignment_modules
        self.modality_tokens = modality_tokens
        self.diffusion_pipes = dict(pipelines or {})
        self.max_context_len = max_context_len
        self.using_lora = using_lora
        self.device = llm.device
        self.decode_modality = dict(IMAGE=self.decode_image, VIDEO=self.decode_video, AUDIO=self.decode_audio)

    # ------------------------------------------------------------------ helpers shared with the free variant
    def embed_tokens(self, ids, using_lora=False):
        return self.llama_model.embed_tokens(ids.to(self.device))

    def get_llm_text_res(self, string, modality):
        return routing.get_llm_text_res(string, modality)

    def get_llm_text_modality(self, string, modalities):
        return routing.get_llm_text_modality(string, modalities)

    @staticmethod
    def split_placeholder(string: str) -> List[str]:   # spider.py:725-740
        out, start = [], 0
        for m in re.finditer(r"<[A-Z]+-Placeholder>", string):
            out.append(string[start:m.start()]); out.append(m.group()); start = m.end()
        out.append(string[start:])
        return out

    def _ids(self, text: str) -> torch.Tensor:
        return self.llama_tokenizer(text, return_tensors="pt", add_special_tokens=False).input_ids.to(self.device)

    # ------------------------------------------------------------------ spider.py:1623-1673
    def prepare_generation_embedding(self, samples):
        embeds = []
        for idx, question in enumerate(samples["Question"]):
            splits = self.split_placeholder(question)
            splits.insert(0, "[INPUT]")
            splits.append(samples["TaskPrompt"][idx])
            if "SystemPrompt" in samples:
                splits.append(samples["SystemPrompt"][idx])
            parts = []
            for s in splits:
                if "Placeholder" in s:
                    raise NotImplementedError("multimodal input placeholders need the input-side encoders (SURVEY 8f N4)")
                parts.append(self.embed_tokens(self._ids(s)))
            embeds.append(torch.cat(parts, dim=1))
        lens = [e.shape[1] for e in embeds]
        max_length = min(max(lens), self.max_context_len)
        pad = self.embed_tokens(torch.tensor([[self.llama_tokenizer.pad_token_id]], device=self.device))[0, 0]
        wrapped = pad.expand(len(lens), max_length, -1).clone()
        atts = torch.zeros(len(lens), max_length, dtype=torch.int32, device=self.device)
        for i, e in enumerate(embeds):          # left padding (spider.py:1658-1661)
            n = min(lens[i], self.max_context_len)
            wrapped[i, -n:] = e[0, :n]
            atts[i, -n:] = 1
        bos = self.embed_tokens(torch.full((len(lens), 1), self.llama_tokenizer.bos_token_id, dtype=torch.int64, device=self.device))
        return torch.cat([bos, wrapped], dim=1), torch.cat([atts[:, :1], atts], dim=1)

    # ------------------------------------------------------------------ spider.py:1413-1463
    def preparing_output_embeds_infer(self, samples, outputs, modality=None, targets=None, modality_i=0):
        if modality is None:
            modality = samples["TaskPrompt"][0][1:-1]
        begin_id, end_id = self._ids(f"<{modality}>"), self._ids(f"</{modality}>")
        if targets is None:
            targets = outputs.sequences[0][1:]
        targets = targets.to(self.device)
        start_pos = (targets == begin_id).nonzero(as_tuple=False)[:, 1].tolist()
        end_pos = (targets == end_id).nonzero(as_tuple=False)[:, 1].tolist()
        hidden, inputs, hidden_text, inputs_text = [], [], [], []
        if modality in self.output_alignment_modules:
            n = self.modality_tokens[modality]
            e, s = end_pos[modality_i], start_pos[modality_i]
            for layer_idx in self.output_alignment_modules[modality]["alignment_layer"]:
                hidden.append(torch.cat([st[layer_idx] for st in outputs.hidden_states[e - n:e]], dim=1))
                inputs.append(self.embed_tokens(targets[e - n:e].unsqueeze(0)))
                hidden_text.append(torch.cat([st[layer_idx] for st in outputs.hidden_states[s + 1:e - n]], dim=1))
                inputs_text.append(self.embed_tokens(targets[s + 1:e - n].unsqueeze(0)))
        return modality, hidden, inputs, hidden_text, inputs_text

    # ------------------------------------------------------------------ decoders (spider.py:346-520)
    def _project(self, hidden_list, input_list, modality):
        proj = None
        if self.output_alignment_MoE_mode is None:      # per-modality TextFcLayer stacks (spider.py:347-351,464-468,502-506)
            assert modality in self.alignment_projs, f"{modality} alignment projections do not exist !"
            projs = self.alignment_projs[modality]
        else:
            projs = self.alignment_projs
        for layer_idx, fc_layer in enumerate(projs):
            h = ops.add(hidden_list[layer_idx].to(BF16).contiguous(), input_list[layer_idx].to(BF16).contiguous())
            p = fc_layer(h, modality=modality)
            proj = p if proj is None else ops.add(proj, p)
        return proj

    def _generate_media(self, modality, samples, proj, return_embeds_only, caption_keys, **call_kwargs):
        if return_embeds_only:
            return proj
        pipe = self.diffusion_pipes.get(modality)
        if pipe is None:
            print(f"no {modality.lower()} generation model.")
            return None
        hidden_embeds_scale = 0.1
        for key in caption_keys:
            if key in samples:
                cond = pipe(samples[key], return_prompts_only=True).detach().to(self.device).to(BF16)
                p, c = torch.broadcast_tensors(proj, cond)       # the reference's `a * proj + b * cond` broadcasts the same way
                proj = ops.axpby(p.contiguous(), c.contiguous(), hidden_embeds_scale, 1 - hidden_embeds_scale)
                break
        return pipe(prompt_embeds=proj, **call_kwargs)

    def decode_image(self, samples, hidden_list, input_list, hidden_text_list, input_text_list, return_embeds_only=True,
                     guidance_scale=7.5, num_inference_steps=40):
        out = self._generate_media("IMAGE", samples, self._project(hidden_list, input_list, "IMAGE"), return_embeds_only,
                                   ("Caption", "llm_text_res"), guidance_scale=guidance_scale, num_inference_steps=num_inference_steps)
        return out if (return_embeds_only or out is None) else out.images

    def decode_video(self, samples, hidden_list, input_list, hidden_text_list, input_text_list, return_embeds_only=True,
                     guidance_scale=7.5, num_inference_steps=40, height=320, width=576, num_frames=16):
        out = self._generate_media("VIDEO", samples, self._project(hidden_list, input_list, "VIDEO"), return_embeds_only,
                                   ("llm_text_res",), guidance_scale=guidance_scale, num_inference_steps=num_inference_steps,
                                   height=height, width=width, num_frames=num_frames)
        return out if (return_embeds_only or out is None) else out.frames

    def decode_audio(self, samples, hidden_list, input_list, hidden_text_list, input_text_list, return_embeds_only=True,
                     guidance_scale=7.5, num_inference_steps=40, audio_length_in_s=5.0):
        out = self._generate_media("AUDIO", samples, self._project(hidden_list, input_list, "AUDIO"), return_embeds_only,
                                   ("llm_text_res",), guidance_scale=guidance_scale, num_inference_steps=num_inference_steps,
                                   audio_length_in_s=audio_length_in_s)
        return out if (return_embeds_only or out is None) else out.audios

    # ------------------------------------------------------------------ spider.py:1526-1621
    def _decode_modality_captions(self, samples, outputs, output_texts, modality, predictions, predictions_text):
        for modality_i, llm_text_res in enumerate(self.get_llm_text_res(output_texts, modality)):
            samples["llm_text_res"] = [llm_text_res]
            modality, h, i_, ht, it = self.preparing_output_embeds_infer(samples, outputs, modality=modality, targets=None,
                                                                         modality_i=modality_i)
            predictions_text[modality].append(llm_text_res)
            preds = self.decode_modality[modality](samples, h, i_, ht, it, return_embeds_only=False)
            if preds is None:
                continue
            predictions[modality].append(preds if modality == "VIDEO" else preds[0])

    def decode_outputs(self, samples, outputs, answers, predictions, predictions_text):
        for i in range(outputs.sequences.shape[0]):
            output_texts = self.llama_tokenizer.decode(outputs.sequences[i], skip_special_tokens=True)
            modality = samples["TaskPrompt"][0][1:-1]
            if modality in self.decode_modality:
                self._decode_modality_captions(samples, outputs, output_texts, modality, predictions, predictions_text)
            elif modality in ("SMARTMULTIMODAL", "SPECIFICMULTIMODAL"):
                for m in self.get_llm_text_modality(output_texts, self.decode_modality.keys()):
                    self._decode_modality_captions(samples, outputs, output_texts, m, predictions, predictions_text)
            elif modality == "IMAGESTORY":
                predictions_text[modality].append(output_texts)
            answers.append(output_texts)
        return answers, predictions, predictions_text

    @torch.no_grad()
    def generate(self, samples, answers, predictions, predictions_text, stop_words_ids=None, num_beams=1, min_length=1,
                 top_p=0.9, repetition_penalty=1, length_penalty=1, temperature=1, do_sample=False):
        inputs_embeds, attention_mask = self.prepare_generation_embedding(samples)
        end_ids = self._ids("[END]")
        stopping = [StoppingCriteriaSub(stops=[[2], end_ids[0].tolist()])]     # EOS id 2 or `[END]` (spider.py:1485-1489)
        outputs = self.llama_model.generate(inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                                            max_new_tokens=self.max_context_len, num_beams=num_beams, do_sample=do_sample,
                                            use_cache=True, stopping_criteria=stopping, output_hidden_states=True,
                                            return_dict_in_generate=True, output_attentions=True)
        return self.decode_outputs(samples, outputs, answers, predictions, predictions_text)
