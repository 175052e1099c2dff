"""hipGraph capture of a fixed-shape forward.

The codec forward is ~60 short launches issued from Python; eager issue leaves
~1 ms of launch gaps per step at batch 32.  ``GraphedForward`` captures one call
of ``fn(static_input)`` into a HIP graph (``torch.cuda.CUDAGraph`` is used only
as the capture/replay front-end; every node is a libagx kernel) and replays it.
libagx is capture-safe by construction: no allocation, no synchronisation, and
its one-time ``hipFuncSetAttribute`` calls happen during the warm-up calls.
"""
from __future__ import annotations

from typing import Callable

import torch


class GraphedForward:
    def __init__(self, fn: Callable, example: torch.Tensor, warmup: int = 2):
        self.static_in = example.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):       # packs weights, sets kernel attributes, warms the allocator
                fn(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = fn(self.static_in)

    def replay(self):
        """Run the captured forward on whatever ``static_in`` currently holds."""
        self.graph.replay()
        return self.static_out

    def __call__(self, x: torch.Tensor):
        self.static_in.copy_(x)
        return self.replay()
