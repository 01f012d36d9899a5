"""SpiderFree end-to-end caller: drop-in for the `predict` flow of qwen2.5omni_spider_web.py:458-521 (SURVEY.md section 2 row 14,
BASELINE configs[1] / [3]):

    text = processor.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)         (:461)
    inputs = processor(text=text, audios=..., images=..., videos=..., return_tensors="pt", padding=True)  (:465)
    text_ids, audio = model.generate(**inputs, spk=voice, use_audio_in_video=True)                     (:468)
    response = processor.batch_decode(text_ids, skip_special_tokens=True, ...)[0].split("\\n")[-1]      (:470-471)
    ask_info['llm_text_all'] = [extract_answer(response)]                                              (:491-493)
    answers, predictions, predictions_text = spider_decoder_infer(ask_info)                            (:520)

    infer = SpiderFreeInfer(thinker=QwenOmniThinker..., processor=..., decoder_infer=SpiderDecoderInfer(cfg))
    res = infer(messages)              # or infer(inputs=processor_output); res.response / .answers / .predictions / .predictions_text

Kept from the reference: the order of the calls, the last-line rule for the response, `extract_answer` (text after `</think>`), the
`ask_info` keys (`llm_text_all`, `Image_ori_array` when the request carried an image), caller-owned containers created by
`SpiderDecoderInfer`, one sample per Decoders-Controller call (spider_decoder.py:311).

Added (the reference serves one request at a time, start to finish): **two-stream pipelining of consecutive requests**,
`SpiderFreeInfer(pipelined=True)` / `.submit()` / `.pipelined(requests)`. A response is an LLM pass (HBM-bound weight streaming)
followed by a decoder pass (a latency-bound chain of short MFMA kernels); consecutive requests are independent, so the decoder
pass of request k is enqueued on one HIP stream (from a helper host thread) while the LLM pass of request k+1 runs on another. Every
request still gets exactly one LLM pass and one decoder pass, each on kernels and workspaces of its own (`ops.workspace_scope`), and
the results are bit-identical to the one-stream order (tests/test_bench_schedule_gpu.py). Rows of one processor batch
(`padding=True`) are answered together: one batched generate, one `SpiderDecoder.generate_batch`.
"""
from __future__ import annotations

import threading
from typing import Any, Dict, Iterable, Iterator, List, Optional

import numpy as np
import torch

from . import ops, routing


class SpiderFreeResult:
    """What `predict` yields for one request (qwen2.5omni_spider_web.py:472,520): the response line, the Decoders-Controller
    triple, and the generated ids (host, [S + new]) they were decoded from."""

    def __init__(self, response: str, answers: list, predictions: dict, predictions_text: dict, text_ids: torch.Tensor):
        self.response, self.answers, self.predictions, self.predictions_text = response, answers, predictions, predictions_text
        self.text_ids = text_ids

    def __iter__(self):          # (answers, predictions, predictions_text) = infer(...) reads like spider_decoder_infer(ask_info)
        return iter((self.answers, self.predictions, self.predictions_text))


class SpiderFreeInfer:
    def __init__(self, thinker, processor, decoder_infer=None, cfg=None, device="cuda:0", generate_kwargs: Optional[dict] = None,
                 pipelined: bool = False, process_mm_info=None, streams=None):
        """thinker: QwenOmniThinker (`model` of the reference); processor: the checkpoint's Qwen2_5OmniProcessor or an object with its
        three methods (apply_chat_template / __call__ / batch_decode); decoder_infer: SpiderDecoderInfer (built from `cfg` when
        omitted); generate_kwargs: extra arguments of every `thinker.generate` call (the reference passes spk / use_audio_in_video);
        process_mm_info: the `qwen_omni_utils.process_mm_info` callable (messages, use_audio_in_video) -> (audios, images, videos);
        text-only messages need none. pipelined: `__call__` returns the PREVIOUS request's result (see `submit`)."""
        if decoder_infer is None:
            if cfg is None:
                raise ValueError("SpiderFreeInfer needs a SpiderDecoderInfer or the config to build one from")
            from .spider_decoder import SpiderDecoderInfer
            decoder_infer = SpiderDecoderInfer(cfg)
        self.model, self.processor, self.spider_decoder_infer = thinker, processor, decoder_infer
        self.device = torch.device(device)
        self.generate_kwargs = dict(generate_kwargs or {})
        self.process_mm_info = process_mm_info
        self.is_pipelined = bool(pipelined)
        self._streams = streams          # optional (LLM stream, decoder stream) of the pipelined schedule, e.g. CU-masked ones (tuning aid)
        self._pending = None             # (text_ids on host, responses, images) of the request whose decoder pass comes next
        self._warm = set()               # LLM-pass geometries that have run (and captured their hipGraphs) on one thread already
        self._last_dec = None            # geometry of the most recent decoder pass (the one the decoders' graphs are captured for)
        self.last_pass_ms: Dict[str, float] = {}

    # ------------------------------------------------------------------ request -> processor output (:461-466)
    def build_inputs(self, messages) -> dict:
        text = self.processor.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)
        audios = images = videos = None
        if self.process_mm_info is not None:
            audios, images, videos = self.process_mm_info(messages, True)
        inputs = self.processor(text=text, audios=audios, images=images, videos=videos, return_tensors="pt", padding=True)
        inputs = dict(inputs)
        inputs["_images"] = images
        return inputs

    # ------------------------------------------------------------------ the two passes of a request
    def llm_pass(self, inputs: dict):
        """`model.generate(**inputs)` + `batch_decode` + the last-line rule, on the CURRENT stream; ends with the one device->host copy
        of the generated ids. -> (text_ids [B, S + new] on the host, one response line per row, the request's input images)."""
        inputs = dict(inputs)
        images = inputs.pop("_images", None)
        with ops.workspace_scope("llm"):
            text_ids = self.model.generate(**inputs, **self.generate_kwargs)
        text_ids = text_ids.cpu()
        resp = self.processor.batch_decode(text_ids, skip_special_tokens=True, clean_up_tokenization_spaces=False)
        return text_ids, [r.split("\n")[-1] for r in resp], images

    def decoder_pass(self, text_ids, responses: List[str], images=None) -> List[SpiderFreeResult]:
        """`ask_info` -> Decoders-Controller for every row of the request (qwen2.5omni_spider_web.py:489-520)."""
        asks = []
        for r in responses:
            ask_info: Dict[str, Any] = {"llm_text_all": [routing.extract_answer(r)]}
            if images is not None:     # inputs of the BOX / MASK decoders (:494-518; SAM / Grounding-DINO themselves are out of scope)
                ask_info["Image_ori_array"] = [np.array(images[0])]
            asks.append(ask_info)
        if len(asks) == 1:
            triples = [self.spider_decoder_infer(asks[0])]
        else:
            triples = self.spider_decoder_infer.spider_decoder.generate_batch(asks)
        return [SpiderFreeResult(r, a, p, pt, text_ids[i]) for i, (r, (a, p, pt)) in enumerate(zip(responses, triples))]

    def _inputs_of(self, messages, inputs):
        if (messages is None) == (inputs is None):
            raise ValueError("pass either `messages` or the processor output `inputs`")
        return self.build_inputs(messages) if inputs is None else inputs

    @staticmethod
    def _unbatch(results: List[SpiderFreeResult]):
        return results[0] if len(results) == 1 else results

    # ------------------------------------------------------------------ hipGraph capture needs a quiet process
    # Every engine captures its hipGraphs on first use of a geometry (decode step per batch size, UNet evaluation per latent shape,
    # tower per input grid); stream capture must not see another host thread's allocations. A pass whose geometry has not run yet is
    # therefore executed alone, on the calling thread; only passes of known geometry are overlapped. The diffusion engines keep ONE
    # graph (their static buffers are sized for one CFG batch): a decoder pass is "known" only if it has the geometry of the decoder
    # pass that ran LAST -- any other one re-captures.
    @staticmethod
    def _llm_key(inputs: dict):
        ids = inputs["input_ids"]
        extra = tuple(sorted((k, tuple(v.shape)) for k, v in inputs.items()
                             if isinstance(v, torch.Tensor) and k not in ("input_ids", "attention_mask")))
        return (int(ids.shape[0]), extra)

    @staticmethod
    def _dec_key(pending):
        calls = [tuple(m for m, _ in routing.route_text(routing.extract_answer(r))[2]) for r in pending[1]]
        return (len(calls), tuple(sorted(calls)))

    # ------------------------------------------------------------------ one request start to finish (the reference's schedule)
    @torch.no_grad()
    def predict(self, messages=None, inputs: Optional[dict] = None):
        inputs = self._inputs_of(messages, inputs)
        pending = self.llm_pass(inputs)
        out = self.decoder_pass(*pending)
        self._warm.add(("llm", self._llm_key(inputs)))
        self._last_dec = self._dec_key(pending)
        return self._unbatch(out)

    def __call__(self, messages=None, inputs: Optional[dict] = None):
        if self.is_pipelined:
            return self.submit(messages, inputs)
        return self.predict(messages, inputs)

    # ------------------------------------------------------------------ two-stream pipelining of consecutive requests
    def _two_streams(self):
        if self._streams is None:
            # the decoder pass is a dependent chain of short kernels: its stream gets the higher priority, so its workgroups are
            # dispatched as soon as a CU frees up; the LLM's long weight-streaming grids fill the rest of the chip
            self._streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device, priority=-1))
        return self._streams

    @torch.no_grad()
    def submit(self, messages=None, inputs: Optional[dict] = None):
        """Hand in request k+1, get the finished result of request k (None when the pipeline was empty): the decoder pass of request k
        runs on one HIP stream, enqueued by a helper host thread, while this thread runs the LLM pass of request k+1 on another stream.
        `flush()` returns the last request's result. Passes of a geometry that has not run before are executed one after the other on
        the calling thread (see above), so the pipeline is full from the third request of a kind on."""
        inputs = self._inputs_of(messages, inputs)
        dev = self.device
        lkey = ("llm", self._llm_key(inputs))
        if self._pending is None:                   # pipeline empty: nothing to overlap with
            self._pending = self.llm_pass(inputs)
            self._warm.add(lkey)
            return None
        pending = self._pending
        dkey = self._dec_key(pending)
        if lkey not in self._warm or dkey != self._last_dec:
            out = self.decoder_pass(*pending)
            self._last_dec = dkey
            self._pending = self.llm_pass(inputs)
            self._warm.add(lkey)
            self.last_pass_ms = {}
            return self._unbatch(out)
        gpu = dev.type == "cuda"                    # (on a CPU device the two passes are simply two host threads: host-logic tests)
        import contextlib
        if gpu:
            sL, sU = self._two_streams()
            cur = torch.cuda.current_stream(dev)
            sU.wait_stream(cur)
            sL.wait_stream(cur)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        on = (lambda st: torch.cuda.stream(st)) if gpu else (lambda st: contextlib.nullcontext())
        mark = (lambda i, st: ev[i].record(st)) if gpu else (lambda i, st: None)
        if not gpu:
            sL = sU = None
        box = {}

        def _dec():
            try:
                if gpu:
                    torch.cuda.set_device(dev)
                with on(sU):
                    mark(0, sU)
                    box["out"] = self.decoder_pass(*pending)
                    mark(1, sU)
            except BaseException as e:              # surfaced on the calling thread below
                box["err"] = e

        th = threading.Thread(target=_dec, name="spider-decoder-enqueue")
        th.start()
        try:
            with on(sL):                            # the LLM pass's ~22 k launches fill its hardware queue: its enqueue blocks this
                mark(2, sL)                         # thread for most of the pass, which is why the decoder has a thread of its own
                self._pending = self.llm_pass(inputs)
                mark(3, sL)
        finally:
            th.join()
        if "err" in box:
            raise box["err"]
        self.last_pass_ms = {"overlapped": True}
        if gpu:
            sU.synchronize()
            sL.synchronize()
            self.last_pass_ms = {"decoder_pass_ms": round(ev[0].elapsed_time(ev[1]), 1), "llm_pass_ms": round(ev[2].elapsed_time(ev[3]), 1)}
        return self._unbatch(box["out"])

    @torch.no_grad()
    def flush(self):
        """Decoder pass of the last submitted request (nothing left to overlap it with)."""
        if self._pending is None:
            return None
        pending, self._pending = self._pending, None
        out = self.decoder_pass(*pending)
        self._last_dec = self._dec_key(pending)
        return self._unbatch(out)

    def pipelined(self, requests: Iterable) -> Iterator:
        """Results of `requests` (messages lists or processor-output dicts) in order, consecutive requests overlapped."""
        for rq in requests:
            r = self.submit(inputs=rq) if isinstance(rq, dict) else self.submit(messages=rq)
            if r is not None:
                yield r
        r = self.flush()
        if r is not None:
            yield r
