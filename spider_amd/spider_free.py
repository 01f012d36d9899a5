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
                 pipelined: bool = False, process_mm_info=None, streams=None, depth: int = 2, mode: str = "spider_free_qwen",
                 story_pipe=None, story_kwargs: Optional[dict] = None, mask_box_inputs=None):
        """thinker: QwenOmniThinker (`model` of the reference); processor: the checkpoint's Qwen2_5OmniProcessor or an object with its
        three methods (apply_chat_template / __call__ / batch_decode); decoder_infer: SpiderDecoderInfer (built from `cfg` when
        omitted); generate_kwargs: extra arguments of every `thinker.generate` call (the reference passes spk / use_audio_in_video);
        process_mm_info: the `qwen_omni_utils.process_mm_info` callable (messages, use_audio_in_video) -> (audios, images, videos);
        text-only messages need none. pipelined: `__call__` returns an EARLIER request's result (see `submit`). depth: requests in
        flight under `submit` -- 2 (default): [LLM pass of k+1 | decoder pass of k]; 3: [decode loop of k+1 | decoder pass of k, then the
        prompt pass of k+2 into a staging KV cache set] (the thinker must offer prefill_begin / adopt / decode_finish). Measured on
        MI355X: depth 3 moves the prompt pass off the LLM stream (LLM pass 510 -> 461 ms) but the step stays at ~514 ms."""
        if depth not in (2, 3):
            raise ValueError("depth must be 2 or 3")
        if mode not in ("spider_free_qwen", "spider_story_free_qwen"):
            raise ValueError("mode must be 'spider_free_qwen' or 'spider_story_free_qwen' (MODEL_NAME of qwen2.5omni_spider_web.py:476,489)")
        # mode = the reference's MODEL_NAME switch: "spider_free_qwen" -> Decoders-Controller (:489-520); "spider_story_free_qwen" ->
        # extract_story_elements + story_generation straight from the response (:476-488; story_pipe = init_story_generation(...)).
        # mask_box_inputs: optional callable(first input image as np.ndarray) -> {"IMAGE_SAM": [...], "Meta_info": {...}}: the SAM-side
        # preprocessing of :497-518 is the caller's (SAM / Grounding-DINO are out of scope); its keys are forwarded in ask_info.
        self.mode, self.story_diffusion, self.story_kwargs = mode, story_pipe, dict(story_kwargs or {})
        self.mask_box_inputs = mask_box_inputs
        if mode == "spider_story_free_qwen" and decoder_infer is None and cfg is None:
            decoder_infer = False          # the story branch needs no Decoders-Controller
        if depth == 3 and not all(hasattr(thinker, m) for m in ("prefill_begin", "adopt", "decode_finish")):
            raise ValueError("depth=3 needs a thinker with prefill_begin / adopt / decode_finish (QwenOmniThinker)")
        self.depth = depth
        if decoder_infer is False:
            decoder_infer = None
        elif decoder_infer is None:
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
        self._prefilled = None           # depth 3: (handle, images, llm key, cache set) of the request whose decode loop comes next
        self._warm = set()               # LLM-pass geometries that have run (and captured their hipGraphs) on one thread already
        self._last_dec = None            # geometry of the most recent decoder pass (the one the decoders' graphs are captured for)
        self.last_pass_ms: Dict[str, float] = {}
        self.host_ms: Dict[str, float] = {}     # host wall time of the pieces of the most recent passes (written by the passes' own threads)

    # ------------------------------------------------------------------ request -> processor output (:461-466)
    def build_inputs(self, messages) -> dict:
        text = self.processor.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)
        audios = images = videos = None
        if self.process_mm_info is not None:
            audios, images, videos = self.process_mm_info(messages, True)
        inputs = self.processor(text=text, audios=audios, images=images, videos=videos, return_tensors="pt", padding=True)
        inputs = dict(inputs)
        inputs["_images"] = [images[0] if images else None]        # per row: the FIRST input image (qwen2.5omni_spider_web.py:495)
        return inputs

    @staticmethod
    def _n_images(messages) -> int:
        """image items of one conversation, in the order process_mm_info walks them"""
        n = 0
        for m in messages:
            c = m.get("content") if isinstance(m, dict) else None
            if isinstance(c, (list, tuple)):
                # qwen_omni_utils.extract_vision_info takes an item as an image when it carries an "image" / "image_url" key or says so
                # in its "type"
                n += sum(1 for it in c if isinstance(it, dict) and
                         ("image" in it or "image_url" in it or it.get("type") in ("image", "image_url")))
        return n

    def build_inputs_batch(self, conversations: List) -> dict:
        """Several conversations as ONE request: one chat-template text per conversation, one processor call with padding. Batched
        generation needs LEFT padding (the prompt must end where generation starts): a processor whose tokenizer pads on the right is
        switched to the left, as HF's `generate` asks for. The rows are then answered together -- one batched LLM pass (the decode
        weight stream serves every row) and one batched decoder pass -- which is what lifts responses per second on one GPU from
        1.95 (one request at a time) to 3.05 / 4.37 / 5.54 at 2 / 4 / 8 rows."""
        texts = [self.processor.apply_chat_template(m, add_generation_prompt=True, tokenize=False) for m in conversations]
        audios = images = videos = None
        if self.process_mm_info is not None:
            audios, images, videos = self.process_mm_info(conversations, True)
        tok = getattr(self.processor, "tokenizer", None)
        if tok is not None and getattr(tok, "padding_side", "left") != "left":
            tok.padding_side = "left"
        inputs = dict(self.processor(text=texts, audios=audios, images=images, videos=videos, return_tensors="pt", padding=True))
        # `images` is the flat list over all conversations: row i owns the next _n_images(conversation i) of them
        counts = [self._n_images(conv) for conv in conversations]
        if images is not None and sum(counts) != len(images):
            raise ValueError(f"the conversations hold {sum(counts)} image items but process_mm_info returned {len(images)} images: "
                             "rows and images would drift apart")
        per_row, k = [], 0
        for n in counts:
            per_row.append(images[k] if (images and n > 0) else None)
            k += n
        inputs["_images"] = per_row
        return inputs

    # ------------------------------------------------------------------ the two passes of a request
    def llm_pass(self, inputs: dict):
        """`model.generate(**inputs)` + `batch_decode` + the last-line rule, on the CURRENT stream; ends with the one device->host copy
        of the generated ids. -> (text_ids [B, S + new] on the host, one response line per row, the request's input images)."""
        import time
        inputs = dict(inputs)
        images = inputs.pop("_images", None)
        t0 = time.perf_counter()
        with ops.workspace_scope("llm"):
            text_ids = self.model.generate(**inputs, **self.generate_kwargs)
        text_ids = text_ids.cpu()
        t1 = time.perf_counter()
        resp = self.processor.batch_decode(text_ids, skip_special_tokens=True, clean_up_tokenization_spaces=False)
        # host wall time of the two halves of the pass (generate ends with the device -> host copy of the ids): read by bench.py
        self.host_ms["llm_generate_host_ms"] = round((t1 - t0) * 1e3, 2)
        self.host_ms["llm_batch_decode_host_ms"] = round((time.perf_counter() - t1) * 1e3, 2)
        return text_ids, [r.split("\n")[-1] for r in resp], images

    def prefill_pass(self, inputs: dict, cache_set: int):
        """depth 3, first half of the LLM pass: towers + prompt pass into KV cache set `cache_set`, enqueued on the current stream."""
        inputs = dict(inputs)
        images = inputs.pop("_images", None)
        with ops.workspace_scope("llm_prefill"):
            handle = self.model.prefill_begin(**inputs, **self.generate_kwargs, cache_set=cache_set)
        return handle, images

    def _adopt(self, pre):
        """Every decode loop runs from KV cache set 0 (ONE decode graph in the process): a request whose prompt pass filled the staging
        set is moved there first -- device copies of its prompt K / V rows and cursors (~0.1 ms) on the CURRENT stream, i.e. ordered
        before everything this step enqueues on either stream (the new prompt pass overwrites the staging set later in the step)."""
        if pre is None:
            return pre
        return (self.model.adopt(pre[0], 0),) + tuple(pre[1:])

    def decode_pass(self, handle, images):
        """depth 3, second half of the LLM pass: decode loop + `batch_decode` + the last-line rule (ends with the device->host copy)."""
        with ops.workspace_scope("llm"):
            text_ids = self.model.decode_finish(handle)
        text_ids = text_ids.cpu()
        resp = self.processor.batch_decode(text_ids, skip_special_tokens=True, clean_up_tokenization_spaces=False)
        return text_ids, [r.split("\n")[-1] for r in resp], images

    def decoder_pass(self, text_ids, responses: List[str], images=None) -> List[SpiderFreeResult]:
        """`ask_info` -> Decoders-Controller for every row of the request (qwen2.5omni_spider_web.py:489-520)."""
        import time
        t_dec0 = time.perf_counter()
        if self.mode == "spider_story_free_qwen":
            return [self._story_result(r, text_ids[i]) for i, r in enumerate(responses)]
        asks = []
        for i, r in enumerate(responses):
            ask_info: Dict[str, Any] = {"llm_text_all": [routing.extract_answer(r)]}
            img = images[i] if (isinstance(images, (list, tuple)) and i < len(images)) else None
            if img is not None:        # inputs of the BOX / MASK decoders (:494-518; SAM / Grounding-DINO themselves are out of scope)
                arr = np.array(img)
                ask_info["Image_ori_array"] = [arr]
                if self.mask_box_inputs is not None:        # the caller's SAM-side preprocessing: IMAGE_SAM, Meta_info (:505-519)
                    extra = self.mask_box_inputs(arr) or {}
                    for k in ("IMAGE_SAM", "Meta_info"):
                        if k in extra:
                            ask_info[k] = extra[k]
            asks.append(ask_info)
        if len(asks) == 1:
            triples = [self.spider_decoder_infer(asks[0])]
        else:
            triples = self.spider_decoder_infer.spider_decoder.generate_batch(asks)
        self.host_ms["decoder_pass_host_ms"] = round((time.perf_counter() - t_dec0) * 1e3, 2)
        return [SpiderFreeResult(r, a, p, pt, text_ids[i]) for i, (r, (a, p, pt)) in enumerate(zip(responses, triples))]

    def _story_result(self, response: str, ids) -> SpiderFreeResult:
        """the `spider_story_free_qwen` branch of predict (qwen2.5omni_spider_web.py:476-488): story elements from the response, then
        story_generation when all three parsed; the reference's error line otherwise. Containers as SpiderStoryFreeInfer fills them."""
        answers, predictions, predictions_text = routing.new_outputs()
        answers.append(response)
        predictions_text["IMAGESTORY"].append(response)
        general_prompt, prompt_array, style_name = routing.extract_story_elements(response)
        if self.story_diffusion is not None and general_prompt and prompt_array and isinstance(prompt_array, list) \
                and len(prompt_array) > 0 and style_name:
            from .story import story_generation
            preds = story_generation(self.story_diffusion, general_prompt=general_prompt, prompt_array=prompt_array, style_name=style_name,
                                     **self.story_kwargs)
            predictions["IMAGESTORY"].append(preds)
            predictions_text["IMAGESTORY_prompts"].append(prompt_array)
        else:
            print("Error: One or more required inputs for story_generation are empty!")
        return SpiderFreeResult(response, answers, predictions, predictions_text, ids)

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

    def _llm_cold(self, inputs: dict, lkey, cache_set: int = 0, decode: bool = True) -> bool:
        """would this LLM pass capture a hipGraph (tower per grid_thw / audio lengths, decode step per row count) -- i.e. must it run
        alone? The engines answer from their own caches (values, evictions and resets included); a thinker without `would_capture`
        falls back to the shapes seen before."""
        if lkey not in self._warm:           # never run by THIS caller (or reset_warm() since): alone, whatever the engines hold
            return True
        wc = getattr(self.model, "would_capture", None)
        if wc is None:
            return False
        # the key the real pass uses: generate(**inputs, **generate_kwargs) -- output_hidden_states / return_logits pick the decode graph
        kw = {k: v for k, v in inputs.items() if not k.startswith("_")}
        kw.update({k: v for k, v in self.generate_kwargs.items() if k not in ("cache_set", "decode")})
        return bool(wc(cache_set=cache_set, decode=decode, **kw))

    def _dec_key(self, pending):
        if self.mode == "spider_story_free_qwen":
            return ("story", len(pending[1]))
        calls = [tuple(m for m, _ in routing.route_text(routing.extract_answer(r))[2]) for r in pending[1]]
        return (len(calls), tuple(sorted(calls)))

    # ------------------------------------------------------------------ one request start to finish (the reference's schedule)
    @torch.no_grad()
    def predict(self, messages=None, inputs: Optional[dict] = None):
        if self._prefilled is not None:
            raise RuntimeError("a prefilled request is in flight (depth-3 pipelining): flush() before calling predict()")
        inputs = self._inputs_of(messages, inputs)
        gpu = self.device.type == "cuda"
        if gpu:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        pending = self.llm_pass(inputs)
        if gpu:
            ev[1].record()
        out = self.decoder_pass(*pending)
        if gpu:
            ev[2].record()
            ev[2].synchronize()
            self.last_pass_ms = {"llm_pass_ms": round(ev[0].elapsed_time(ev[1]), 1), "decoder_pass_ms": round(ev[1].elapsed_time(ev[2]), 1),
                                 "overlapped": False}
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
            import os
            pl, pu = (int(v) for v in os.environ.get("SPIDER_STREAM_PRIO", "0,-1").split(","))      # tuning aid: "LLM,decoder" priorities
            self._streams = (torch.cuda.Stream(device=self.device, priority=pl), torch.cuda.Stream(device=self.device, priority=pu))
        return self._streams

    @torch.no_grad()
    def submit(self, messages=None, inputs: Optional[dict] = None):
        """Hand in request k+1, get the finished result of request k (None when the pipeline was empty): the decoder pass of request k
        runs on one HIP stream, enqueued by a helper host thread, while this thread runs the LLM pass of request k+1 on another stream.
        `flush()` returns the last request's result. Passes of a geometry that has not run before are executed one after the other on
        the calling thread (see above), so the pipeline is full from the third request of a kind on."""
        inputs = self._inputs_of(messages, inputs)
        if self.depth == 3:
            return self._submit3(inputs)
        lkey = ("llm", self._llm_key(inputs))
        if self._pending is None:                   # pipeline empty: nothing to overlap with
            self._pending = self.llm_pass(inputs)
            self._warm.add(lkey)
            return None
        pending = self._pending
        dkey = self._dec_key(pending)
        if self._llm_cold(inputs, lkey) or dkey != self._last_dec or self.mode == "spider_story_free_qwen":
            # one pass after the other on this thread, with the state rules of the overlapped step: the old request leaves the pipeline
            # whether its decoder pass succeeds or not (a request that fails deterministically must not wedge every later submit, and a
            # request that has been decoded must never be decoded again), the new request's LLM pass runs either way, errors are raised last
            self._pending = None
            self.last_pass_ms = {}
            out, err = self._try(lambda: self.decoder_pass(*pending))
            if err is None:
                self._last_dec = dkey
            newpend, lerr = self._try(lambda: self.llm_pass(inputs))
            if lerr is None:
                self._pending = newpend
                self._warm.add(lkey)
            self._raise_step_errors(lerr, err, out)
            return self._unbatch(out)
        out, newpend, err, lerr = self._overlap(lambda: self.decoder_pass(*pending), lambda: self.llm_pass(inputs))
        # the old request's decoder pass has run (or failed) either way: it must never be decoded again. The NEW request's LLM result
        # survives a failed decoder pass of the old one; a failed LLM pass leaves the pipeline empty.
        self._pending = newpend
        self._raise_step_errors(lerr, err, out)
        return self._unbatch(out)

    def _overlap(self, on_u, on_l):
        """run on_u() on the decoder stream from a helper host thread and on_l() on the LLM stream from this thread; both finished
        (device included) on return. -> (result of on_u, result of on_l, the helper thread's exception or None, this thread's exception
        or None): a failed pass must not lose the result computed beside it, and must not leave a request in the pipeline that has
        already been decoded -- the caller stores the new state first, then raises. The LLM pass's ~22 k launches fill its hardware queue, so its
        enqueue blocks the enqueueing thread for most of the pass: that is why the decoder pass has a thread of its own. (On a CPU
        device the two passes are simply two host threads: host-logic tests.)"""
        dev = self.device
        gpu = dev.type == "cuda"
        import contextlib
        sL = sU = None
        if gpu:
            sL, sU = self._two_streams()
            cur = torch.cuda.current_stream(dev)
            sU.wait_stream(cur)
            sL.wait_stream(cur)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        on = (lambda st: torch.cuda.stream(st)) if gpu else (lambda st: contextlib.nullcontext())
        mark = (lambda i, st: ev[i].record(st)) if gpu else (lambda i, st: None)
        box = {}

        def _u():
            try:
                if gpu:
                    torch.cuda.set_device(dev)
                with on(sU):
                    mark(0, sU)
                    box["u"] = on_u()
                    mark(1, sU)
            except BaseException as e:              # surfaced on the calling thread below
                box["err"] = e

        th = threading.Thread(target=_u, name="spider-decoder-enqueue")
        th.start()
        try:
            with on(sL):
                mark(2, sL)
                box["l"] = on_l()
                mark(3, sL)
        except BaseException as e:                  # handed back like the helper's: the caller advances its state, then raises
            box["lerr"] = e
        finally:
            th.join()
            if gpu:                                   # nothing of this step is in flight when we return, error or not
                sU.synchronize()
                sL.synchronize()
        self.last_pass_ms = {"overlapped": True}
        if gpu and "err" not in box and "lerr" not in box:
            self.last_pass_ms = {"decoder_pass_ms": round(ev[0].elapsed_time(ev[1]), 1), "llm_pass_ms": round(ev[2].elapsed_time(ev[3]), 1)}
        return box.get("u"), box.get("l"), box.get("err"), box.get("lerr")

    @staticmethod
    def _try(fn):
        """-> (fn(), None) or (None, the exception it raised): the passes of a step run to the end before anything is raised"""
        try:
            return fn(), None
        except BaseException as e:
            return None, e

    def _raise_step_errors(self, lerr, err, out=None):
        """raise what a step's two passes raised, AFTER the caller has advanced the pipeline state: this thread's exception first (the
        helper's chained onto it), else the helper's. A result that was computed beside the failure rides on the exception
        (`.spider_result`) instead of being lost."""
        first = lerr if lerr is not None else err
        if first is None:
            return
        if lerr is not None and err is not None and lerr.__cause__ is None:
            lerr.__cause__ = err
        if out is not None:
            try:
                first.spider_result = self._unbatch(out)
            except Exception:
                pass
        raise first

    # ------------------------------------------------------------------ depth 3: the prompt pass rides on the decoder stream
    def _submit3(self, inputs: dict):
        """depth 3. Step k runs, concurrently: stream L / this thread: the decode loop of request k+1 (prefilled one step ago);
        stream U / helper thread: the decoder pass of request k, THEN the prompt pass of the new request k+2 into the STAGING KV cache
        set (request k+1 was moved from there into set 0 at the top of the step: `_adopt`) -- the decoder stream has slack (the decode
        loop is the longer half), so the prompt pass leaves the critical stream. A request's result comes back two submits after it was
        handed in; passes whose hipGraphs do not exist yet run alone (as in depth 2)."""
        lkey = ("llm", self._llm_key(inputs))
        cset = 1        # the prompt pass always fills the STAGING set; decode loops run from set 0 (decode_pass adopts the request)
        pre, pend = self._adopt(self._prefilled), self._pending
        B_new = int(inputs["input_ids"].shape[0])
        if pre is None and pend is None:            # empty pipeline: the new request's prompt pass only
            self._prefilled = (*self.prefill_pass(inputs, cset), lkey, cset, B_new)
            self._warm.add(lkey)
            return None
        dgk = ("dg", pre[4], 0) if pre is not None else None           # decode graph of (rows, cache set 0)
        dkey = self._dec_key(pend) if pend is not None else None
        # (the new request's PROMPT pass captures tower graphs at most: decode loops always run from set 0, whose graph `dgk` names)
        warm = (pre is not None and pend is not None and not self._llm_cold(inputs, lkey, cset, decode=False) and dgk in self._warm and
                dkey == self._last_dec and self.mode != "spider_story_free_qwen")
        if not warm:                                 # some graph of this step does not exist yet: one pass after the other, one thread
            # (state rules of the overlapped step: both old requests leave their slots before their passes run -- a decoded or adopted
            # request is never run again, a request that fails deterministically cannot wedge the pipeline -- every pass runs, errors last)
            self._pending, self._prefilled = None, None
            self.last_pass_ms = {}
            out = err = lerr = perr = None
            if pend is not None:
                out, err = self._try(lambda: self.decoder_pass(*pend))
                if err is None:
                    self._last_dec = dkey
            if pre is not None:
                newpend, lerr = self._try(lambda: self.decode_pass(pre[0], pre[1]))
                if lerr is None:
                    self._pending = newpend
                    self._warm.add(dgk)
            newpre, perr = self._try(lambda: self.prefill_pass(inputs, cset))
            if perr is None:
                self._prefilled = (*newpre, lkey, cset, B_new)
                self._warm.add(lkey)
            first = lerr if lerr is not None else perr
            if first is not None and perr is not None and first is not perr and first.__cause__ is None:
                first.__cause__ = perr
            self._raise_step_errors(first, err, out)
            return None if out is None else self._unbatch(out)

        box = {}

        def on_u():
            box["res"] = self.decoder_pass(*pend)
            return self.prefill_pass(inputs, cset)

        newpre, newpend, err, lerr = self._overlap(on_u, lambda: self.decode_pass(pre[0], pre[1]))
        # both old requests have been consumed by this step (decoded / adopted into set 0 and run): drop them before anything below can
        # raise, or a later flush() / submit() would decode the adopted request again from a KV set that has been overwritten
        self._pending, self._prefilled = newpend, None
        if newpre is None and lerr is None:   # the helper failed before / inside the new request's prompt pass: run it here
            try:
                newpre = self.prefill_pass(inputs, cset)
            except BaseException as e:
                if err is not None and e.__cause__ is None:
                    e.__cause__ = err
                raise
        if newpre is not None:
            self._prefilled = (*newpre, lkey, cset, B_new)
            self._warm.add(lkey)
        self._raise_step_errors(lerr, err, box.get("res"))
        return self._unbatch(box["res"])

    def reset_warm(self):
        """Forget which pass geometries have run: call after swapping or reloading an engine behind the decoders / the model (a new
        engine has no hipGraphs yet, and a capture must not happen beside another thread's launches). The next passes of every
        geometry run one after the other on the calling thread again, as the first ones did."""
        if self._pending is not None or self._prefilled is not None:
            raise RuntimeError("reset_warm() with requests in flight: flush() first")
        self._warm.clear()
        self._last_dec = None

    @torch.no_grad()
    def flush(self):
        """Drain one request: the decoder pass of the oldest request in flight (nothing left to overlap it with); with depth 3 a
        prefilled request advances to its decode loop. Call until it returns None."""
        if self.depth == 3:
            if self._pending is None and self._prefilled is not None:
                pre, self._prefilled = self._adopt(self._prefilled), None
                self._pending = self.decode_pass(pre[0], pre[1])
                self._warm.add(("dg", pre[4], 0))
            if self._pending is None:
                return None
            pend, self._pending = self._pending, None
            out = self.decoder_pass(*pend)
            self._last_dec = self._dec_key(pend)
            if self._prefilled is not None:
                pre, self._prefilled = self._adopt(self._prefilled), None
                self._pending = self.decode_pass(pre[0], pre[1])
                self._warm.add(("dg", pre[4], 0))
            return self._unbatch(out)
        if self._pending is None:
            return None
        pending, self._pending = self._pending, None
        out = self.decoder_pass(*pending)
        self._last_dec = self._dec_key(pending)
        return self._unbatch(out)

    def pipelined(self, requests: Iterable, group: int = 1) -> Iterator:
        """Results of `requests` (messages lists or processor-output dicts) in order, consecutive requests overlapped.
        group > 1: up to `group` consecutive conversations (messages lists) are merged into one batched request
        (`build_inputs_batch`) and answered together; every conversation still gets a result of its own, in request order.
        Processor-output dicts pass through as they are (a dict with several rows yields a list, as `predict` returns it)."""
        if group < 1:
            raise ValueError("group must be >= 1")

        def emit(r, merged):
            # a merged request comes back as a list of per-row results (one row: the bare result): one result per conversation
            if merged:
                yield from (r if isinstance(r, list) else [r])
            else:
                yield r

        merged_flags: List[bool] = []          # FIFO: was the request in flight built by merging conversations?
        bucket: List = []

        def send_bucket():
            convs = list(bucket)
            bucket.clear()
            merged_flags.append(True)
            return self.submit(inputs=self.build_inputs_batch(convs))

        for rq in requests:
            outs = []
            if isinstance(rq, dict) or group == 1:
                if bucket:
                    outs.append(send_bucket())
                merged_flags.append(False)
                outs.append(self.submit(inputs=rq) if isinstance(rq, dict) else self.submit(messages=rq))
            else:
                bucket.append(rq)
                if len(bucket) == group:
                    outs.append(send_bucket())
            for r in outs:
                if r is not None:
                    yield from emit(r, merged_flags.pop(0))
        if bucket:
            r = send_bucket()
            if r is not None:
                yield from emit(r, merged_flags.pop(0))
        while True:
            r = self.flush()
            if r is None:
                break
            yield from emit(r, merged_flags.pop(0))
