"""Weight prefetch into the MI355X Infinity Cache for the latency-bound UNet step (MI355X-side design; the reference has no
counterpart: `self.unet(...)` streams 1.7 GB of fp16 weights cold from HBM at every denoising step, custom_sd.py:634-639).

Why: at CFG batch 2 the SD-v1.5 UNet step is ~330 dependent launches that together use ~5 % of the HBM bandwidth; its 16^2 / 8^2
levels are weight-streaming problems whose K loops run 0.38 us per tile on HBM-cold weights but 0.26 us on weights that sit in the
256 MB memory-side cache (scripts/exp/small_gemm_floor.py, profiles/r04_small_gemm_floor.txt). The weights (1.7 GB) do not fit, but
a reader that runs a bounded WINDOW ahead of the compute chain makes every weight warm by the time its kernel starts, using
bandwidth the chain leaves idle.

How: the engine's forward is traced once (ops.set_weight_hook: the tensors its kernels read, in launch order); under stream capture
the same forward runs with a hook that, before compute launch i is enqueued, forks onto a side stream the prefetch launches of all
weights whose position in the byte stream is at most `window` bytes ahead of launch i (cyclically: the tail of the step fetches
the head of the next one, since the graph is replayed step after step). Fork = an event recorded on the capture stream before
launch i, waited for by the side stream: the reader can never run further ahead than the window, so it cannot evict what it
fetched before use. The side stream rejoins the capture stream at the end of the forward. Results are untouched (the prefetch
kernel only reads); graph replay == eager stays bit-identical.
"""
import os
from typing import Callable, List, Optional

import torch

from . import lib as _lib
from . import ops


def _env_int(name: str, default: int) -> int:
    v = os.environ.get(name)
    return int(v) if v else default


class WeightPrefetcher:
    def __init__(self, device, window_mb: Optional[int] = None, blocks: Optional[int] = None, nt: Optional[bool] = None,
                 min_bytes: int = 256 << 10):
        self.device = torch.device(device)
        self.window = (window_mb if window_mb is not None else _env_int("SPIDER_PREFETCH_WINDOW_MB", 96)) << 20
        self.blocks = blocks if blocks is not None else _env_int("SPIDER_PREFETCH_BLOCKS", 48)
        self.nt = bool(nt if nt is not None else _env_int("SPIDER_PREFETCH_NT", 0))
        self.min_bytes = min_bytes
        self.group = _env_int("SPIDER_PREFETCH_GROUP_MB", 32) << 20
        self.trace: List[torch.Tensor] = []          # weight read by weight-streaming launch i (tensors kept alive here)
        self.gated: List[List[int]] = []             # gated[i] = indices j whose prefetch is forked before launch i
        self.sink = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.bytes_per_step = 0

    # ------------------------------------------------------------------ plan
    def record(self, forward: Callable[[], object]):
        """Run `forward` eagerly and note the weights its kernels read, in launch order; then build the fork schedule."""
        seen: List[torch.Tensor] = []
        prev = ops.set_weight_hook(seen.append)
        try:
            out = forward()
        finally:
            ops.set_weight_hook(prev)
        self.trace = seen
        self._schedule()
        return out

    def _schedule(self):
        n = len(self.trace)
        size = [t.numel() * t.element_size() for t in self.trace]
        cum = [0] * (n + 1)
        for i in range(n):
            cum[i + 1] = cum[i] + size[i]
        total = cum[n]
        self.bytes_per_step = total
        self.gated = [[] for _ in range(n)]
        if n == 0 or total <= self.window:       # everything fits the cache: it stays warm by itself from replay to replay
            return
        last_ptr_pos = {}
        for j in range(n):
            if size[j] < self.min_bytes:
                continue
            ptr = self.trace[j].data_ptr()
            if ptr in last_ptr_pos and cum[j] - last_ptr_pos[ptr] <= self.window:    # read again within the window: still warm
                last_ptr_pos[ptr] = cum[j]
                continue
            last_ptr_pos[ptr] = cum[j]
            start = cum[j] - self.window           # byte position the compute chain must have reached
            if start < 0:
                start += total                      # ... in the previous replay of the graph
            # first launch g whose preceding launches cover `start` bytes (binary search over cum)
            lo, hi = 0, n - 1
            while lo < hi:
                mid = (lo + hi) // 2
                if cum[mid] >= start:
                    hi = mid
                else:
                    lo = mid + 1
            self.gated[lo].append(j)
        # Fork points are expensive inside a hipGraph (a cross-stream edge costs ~10 us on the capture stream's own chain: 159
        # forks made the 5.7 ms step 7.4 ms): merge them until each carries at least `group` bytes; a merged batch forks at the
        # EARLIEST gate of its members, so the reader runs at most window + group ahead.
        if self.group > 0:
            merged = [[] for _ in range(n)]
            # walk the fork points in cyclic order starting at the first one; batch = consecutive fork points
            order = [i for i in range(n) if self.gated[i]]
            head, acc = None, 0
            for i in order:
                if head is None:
                    head, acc = i, 0
                merged[head].extend(self.gated[i])
                acc += sum(size[j] for j in self.gated[i])
                if acc >= self.group:
                    head = None
            self.gated = merged

    # ------------------------------------------------------------------ capture
    def run(self, forward: Callable[[], object]):
        """Run `forward` on the current stream (normally under stream capture) with the prefetch launches forked beside it."""
        if not any(self.gated):
            return forward()
        main = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(main)
        state = {"i": 0}
        n = len(self.trace)

        def hook(t):
            i = state["i"]
            state["i"] = i + 1
            if i >= n or t.data_ptr() != self.trace[i].data_ptr():
                raise RuntimeError("WeightPrefetcher: the forward launched other kernels than the traced one (launch "
                                   f"{i}); record() must see the same geometry and flags as run()")
            js = self.gated[i]
            if not js:
                return
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            for j in js:
                w = self.trace[j]
                _lib.call("spider_prefetch_weights", w.data_ptr(), w.numel() * w.element_size(), self.blocks, int(self.nt),
                          self.sink.data_ptr(), side.cuda_stream)

        prev = ops.set_weight_hook(hook)
        try:
            out = forward()
        finally:
            ops.set_weight_hook(prev)
        if state["i"] != n:
            raise RuntimeError(f"WeightPrefetcher: traced {n} weight-streaming launches, ran {state['i']}")
        main.wait_stream(side)
        return out

    def describe(self) -> dict:
        return {"launches_traced": len(self.trace), "prefetch_launches": sum(len(g) for g in self.gated),
                "weight_bytes_per_step": self.bytes_per_step, "window_bytes": self.window, "blocks": self.blocks, "nt": self.nt,
                "forks": sum(1 for g in self.gated if g), "group_bytes": self.group}
