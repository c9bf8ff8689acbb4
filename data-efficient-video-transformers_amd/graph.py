"""hipGraph capture of a whole training step.

The step of the clip path is a fixed sequence of ~300 kernel launches, a third of
them microsecond-sized (the 33-token temporal encoder); launched one by one from
Python the host cannot keep the GPU fed.  ``capture_step`` records the sequence
once (HIP stream capture through ``torch.cuda.CUDAGraph``: every launch of
libdvt_hip.so on the capturing stream is recorded) and replays it with a single
``hipGraphLaunch`` per step.  Requirements on ``step_fn``: static shapes, inputs
read from fixed tensors, no host synchronisation (no ``.item()``), optimizer state
on the device (``dp.FlatParameters.adamw_step`` keeps its step counter there).
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch


def capture_step(step_fn: Callable[[], torch.Tensor], warmup: int = 3) -> Tuple[Callable[[], None], torch.Tensor]:
    """Returns (replay, static_output).  ``replay()`` re-runs the captured step;
    ``static_output`` is the tensor returned by ``step_fn`` during capture (its
    storage is overwritten by every replay)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, warmup)):     # allocates workspaces, sets kernel attributes
            step_fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step_fn()
    return graph.replay, out
