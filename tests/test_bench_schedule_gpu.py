"""GPU: bench.py's two-stream schedule (the decoder pass of response k beside the LLM pass of response k+1, each in its own
split-K workspace scope) returns bit-identical responses to the one-stream schedule: the two passes share no mutable state."""
import argparse
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_overlapped_schedule_equals_serial(dev):
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.DIFF_DT = torch.float16
    args = argparse.Namespace(llm="qwen25_7b", batch=1, throughput_batch=0, prompt_len=192, new_tokens=12, denoise_steps=5,
                              schedule="overlap", workload="text_image", no_stream32=False)
    resp = bench.Responder(args, dev)
    ref_tok, ref_img = resp.respond_serial()
    ref_tok, ref_img = ref_tok.clone(), ref_img.clone()
    outs = [resp.respond() for _ in range(3)]          # step 0 primes the pipeline, steps 1-2 run both passes concurrently
    torch.cuda.synchronize()
    assert resp.overlap_ms and resp.overlap_ms["decoder_pass_ms"] > 0 and resp.overlap_ms["llm_pass_ms"] > 0
    for tok, img in outs:
        assert torch.equal(tok, ref_tok), "tokens differ between the overlapped and the serial schedule"
        assert torch.equal(img, ref_img), "image differs between the overlapped and the serial schedule"
    assert ref_img.dtype == torch.uint8 and tuple(ref_img.shape) == (1, 3, 512, 512) and int(ref_img.max()) > int(ref_img.min())
