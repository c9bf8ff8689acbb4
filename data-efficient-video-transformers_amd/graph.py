"""hipGraph capture of a whole training step.

The step of the clip path is a fixed sequence of ~300 kernel launches, a third of
them microsecond-sized (the 33-token temporal encoder); launched one by one from
Python the host cannot keep the GPU fed.  ``capture_step`` records the sequence
once (HIP stream capture through ``torch.cuda.CUDAGraph``: every launch of
libdvt_hip.so on the capturing stream is recorded) and replays it with a single
``hipGraphLaunch`` per step.  Requirements on ``step_fn``: static shapes, inputs
read from fixed tensors, no host synchronisation (no ``.item()``), optimizer state
on the device (``dp.FlatParameters.adamw_step`` keeps its step counter there).

Data parallel: the gradient buckets' RCCL all-reduces (``dp.Communicator``: ``dvt_comm_allreduce``
on a side stream, forked from and joined to the compute stream through events) are
recorded in the same graph, so the N-GPU step is the same single ``hipGraphLaunch`` as
the one-GPU step.  ``capture_step_segments`` is the form for an exchange that cannot
be captured (gloo on the CPU rehearsal, or an RCCL build that refuses capture): the
step is cut at the exchange into two graphs with the collective launched between them.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch


def capture_step(step_fn: Callable[[], torch.Tensor], warmup: int = 3, flat=None) -> Tuple[Callable[[], None], torch.Tensor]:
    """Returns (replay, static_output).  ``replay()`` re-runs the captured step;
    ``static_output`` is the tensor returned by ``step_fn`` during capture (its
    storage is overwritten by every replay).  ``flat`` (a ``dp.FlatParameters``): its cached packed convolution
    weights are marked stale before the capture, so the one repack launch per step is recorded in the graph even when
    ``step_fn`` does not contain the optimizer step that normally invalidates them."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, warmup)):     # allocates workspaces, sets kernel attributes
            step_fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if flat is not None:
        flat.invalidate_packed()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step_fn()
    return graph.replay, out


def capture_step_segments(fwd_bwd_fn: Callable[[], torch.Tensor], exchange_fn: Callable[[], None],
                          update_fn: Callable[[], None], warmup: int = 3, flat=None) -> Tuple[Callable[[], None], torch.Tensor]:
    """Two graphs around an eagerly launched exchange:  replay() = graph(fwd_bwd) ; exchange_fn() ; graph(update).
    ``fwd_bwd_fn`` must leave the gradient exchange to ``exchange_fn`` (``FlatParameters.finish_backward(exchange=False)``
    then ``FlatParameters.exchange_all()``)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, warmup)):
            fwd_bwd_fn()
            exchange_fn()
            update_fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if flat is not None:
        flat.invalidate_packed()
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        out = fwd_bwd_fn()
    exchange_fn()
    with torch.cuda.graph(g2, pool=g1.pool()):
        update_fn()

    def replay():
        g1.replay()
        exchange_fn()
        g2.replay()

    return replay, out
