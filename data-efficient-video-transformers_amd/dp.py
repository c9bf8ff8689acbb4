"""Flat parameter / gradient storage and data-parallel gradient exchange.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).
The clips of the global batch are independent through the whole forward
(SURVEY section 8e), so data parallelism needs exactly one exchange: the sum of
the parameter gradients.  The reference itself is single-GPU
(``pl.Trainer(gpus=1)``, src/main.py:87); this module is the new MI355X-side
functionality north_star asks for.

Design (MI355X-first, not a DDP clone):
  * all fp32 master parameters live in ONE flat HBM buffer, all gradients in a
    second one; ``param.data`` / ``param.grad`` are views.  The backward kernels
    write weight gradients straight into their slice (``GradSink``): no per-
    parameter gradient tensors, no flatten/copy, no torch accumulate kernels.
  * buckets are contiguous ranges of the gradient buffer, formed in reverse
    registration order (the order in which backward completes them).  When the
    last gradient of a bucket has been written, its all-reduce is enqueued
    asynchronously; RCCL runs it on its own stream, overlapped with the rest of
    backward.  xGMI is point-to-point (7 links x ~153 GB/s), a ring all-reduce is
    bound by one link, so buckets are large (default 32 MiB: ~2 buckets for the
    115 MB of fp32 gradients of the d=512 model) to amortise latency.
  * averaging is folded into the loss gradient (1/world), so the collective is a
    plain sum and no extra pass over the gradients exists.
  * the optimizer step is one fused launch over the flat buffers: AdamW, the
    16-bit compute copy of the updated weights, and the device step counter.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

import ctypes as _C

from . import _lib as L
from . import ops

_ALIGN = 64  # elements: every parameter slice starts 256-byte aligned


class _Enqueued:
    """Handle of a collective enqueued on the communicator's side stream: ``wait()`` makes the CURRENT stream wait for
    it (device-side dependency, no host synchronisation -- capturable in a hipGraph)."""

    __slots__ = ("event",)

    def __init__(self, event):
        self.event = event

    def wait(self) -> None:
        torch.cuda.current_stream().wait_event(self.event)


class Communicator:
    """RCCL communicator behind the C ABI (``dvt_comm_*``, include/dvt_hip.h): the one collective of the path, the SUM
    of the gradient buckets over xGMI.  Collectives are enqueued on a side stream of their own behind an event of the
    compute stream, so a bucket's exchange overlaps the backward kernels that follow it, eagerly and inside a captured
    hipGraph alike (the fork / join through events is what stream capture records)."""

    def __init__(self, world: int, rank: int, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("RCCL unique id must be 128 bytes")
        self.world, self.rank = world, rank
        self._h = _C.c_void_p()
        L.check(L.load().dvt_comm_init(_C.byref(self._h), unique_id, world, rank), "dvt_comm_init")
        self.stream = torch.cuda.Stream()
        # Diagnostic trace (bench.py, eager launches only -- timing events cannot be captured in a hipGraph): when a list,
        # every all_reduce_async appends (bytes on the wire, start event, end event) recorded on the side stream, and
        # ``marks`` holds named timing events of the compute stream (FlatParameters records "backward_end").
        self.trace = None
        self.marks = {}

    @staticmethod
    def unique_id() -> bytes:
        buf = _C.create_string_buffer(128)
        L.check(L.load().dvt_comm_unique_id(buf), "dvt_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, group: Optional[dist.ProcessGroup] = None) -> "Communicator":
        """Bootstrap over an initialised torch.distributed group (only the 128-byte id travels through it)."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(world, rank, box[0])

    def _after_current(self) -> None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)

    def _done(self) -> _Enqueued:
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return _Enqueued(ev)

    def all_reduce_async(self, t: torch.Tensor, via: Optional[torch.Tensor] = None) -> _Enqueued:
        """In-place SUM of ``t`` over the ranks, on the side stream, after everything enqueued so far on the current
        stream.  ``via`` (a 16-bit buffer of the same length): exchange a half-width copy -- cast, all-reduce, cast back,
        all on the side stream (57.7 MB instead of 115 MB over the per-link-bound ring)."""
        assert t.is_cuda and t.is_contiguous()
        lib = L.load()
        self._after_current()
        st = self.stream.cuda_stream
        n = t.numel()
        if self.trace is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
        if via is None:
            L.check(lib.dvt_comm_allreduce(self._h, t.data_ptr(), n, ops._DT[t.dtype], st), "dvt_comm_allreduce")
        else:
            assert via.numel() == n and via.dtype in (torch.bfloat16, torch.float16)
            L.check(lib.dvt_cast(t.data_ptr(), ops._DT[t.dtype], via.data_ptr(), ops._DT[via.dtype], n, st), "dvt_cast")
            L.check(lib.dvt_comm_allreduce(self._h, via.data_ptr(), n, ops._DT[via.dtype], st), "dvt_comm_allreduce")
            L.check(lib.dvt_cast(via.data_ptr(), ops._DT[via.dtype], t.data_ptr(), ops._DT[t.dtype], n, st), "dvt_cast")
        if self.trace is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(self.stream)
            self.trace.append((n * (t if via is None else via).element_size(), e0, e1))
        return self._done()

    def broadcast(self, t: torch.Tensor, root: int = 0) -> None:
        assert t.is_cuda and t.is_contiguous()
        self._after_current()
        L.check(L.load().dvt_comm_broadcast(self._h, t.data_ptr(), t.numel(), ops._DT[t.dtype], root,
                                            self.stream.cuda_stream), "dvt_comm_broadcast")
        self._done().wait()

    def destroy(self) -> None:
        if self._h:
            torch.cuda.synchronize()
            L.check(L.load().dvt_comm_destroy(self._h), "dvt_comm_destroy")
            self._h = _C.c_void_p()


class GradSink:
    """Destination of one parameter's gradient inside the flat gradient buffer.

    Writers (functional.py) follow one protocol: write into ``sink.buf`` with ``accumulate = not sink.fresh``, then
    call ``sink.mark_written()``.  A parameter may be written more than once per step (a head used on two streams, a
    shared encoder).  Under data parallelism its bucket's all-reduce may already be in flight when a later write
    arrives: adding a local gradient to the (partly) reduced sum would be wrong on every rank and races with the
    collective.  Such a *late* write is redirected transparently: ``buf`` / ``fresh`` then name a per-sink side buffer,
    which ``FlatParameters.finish_backward`` all-reduces by itself and adds to the reduced gradient."""

    __slots__ = ("_buf", "_fresh", "bucket", "owner", "index", "unwritten", "_late", "_late_fresh", "late_written",
                 "_zeroed", "_uncounted")

    def __init__(self, buf: torch.Tensor, bucket: int, owner: "FlatParameters", index: int):
        self._buf = buf         # view shaped like the parameter
        self._fresh = True      # True until first written in the current step
        self.bucket = bucket
        self.owner = owner
        self.index = index
        self.unwritten = False  # nobody wrote it in the step that just finished (the optimizers skip it, like grad None)
        self._late = None       # side buffer for writes that arrive after the bucket's all-reduce was launched
        self._late_fresh = True
        self.late_written = False
        self._zeroed = False    # the slice holds zeros written by finish_backward and nobody has written it since
        self._uncounted = False  # this step's bucket count does not wait for it (nobody wrote it in the previous step)

    def _is_late(self) -> bool:
        o = self.owner
        return o._exchanging() and o._launched[self.bucket]

    @property
    def buf(self) -> torch.Tensor:
        if self._is_late():
            if self._late is None:
                self._late = torch.zeros_like(self._buf)
            return self._late
        return self._buf

    @property
    def fresh(self) -> bool:
        return self._late_fresh if self._is_late() else self._fresh

    def mark_written(self) -> None:
        if self._is_late():
            self._late_fresh = False
            self.late_written = True
            return
        if self._fresh:
            self._fresh = False
            self.unwritten = False
            self._zeroed = False
            if not self._uncounted:          # (a parameter the bucket did not wait for: nothing to count down)
                self.owner._on_first_write(self)


def sink_of(p) -> Optional[GradSink]:
    return getattr(p, "_dvt_sink", None)


class FlatParameters:
    def __init__(self, module: torch.nn.Module, *, bucket_mb: float = 32.0,
                 process_group: Optional[dist.ProcessGroup] = None,
                 compute_dtype: Optional[torch.dtype] = torch.bfloat16,
                 comm: Optional[Communicator] = None, grad_reduce_dtype: Optional[torch.dtype] = None):
        """comm: exchange the gradient buckets through the C-ABI RCCL communicator (GPU tensors; capturable in a
        hipGraph) instead of torch.distributed collectives (kept for the CPU / gloo rehearsal of the same logic).
        grad_reduce_dtype: torch.bfloat16 / float16 = half-width exchange of the buckets (needs ``comm``)."""
        seen, params = set(), []
        for p in module.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("module has no trainable parameters")
        dev = params[0].device   # CPU tensors are accepted for the exchange logic only (gloo tests);
        # the fused optimizer / compute-copy kernels need the GPU and raise otherwise.
        self.params: List[torch.nn.Parameter] = params
        self.offsets, total = [], 0
        for p in params:
            self.offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = total
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.compute_dtype = compute_dtype
        self.compute = (torch.zeros(total, dtype=compute_dtype, device=dev)
                        if compute_dtype not in (None, torch.float32) and dev.type == "cuda" else None)
        self.compute_valid = False
        self.group = process_group
        self.comm = comm
        if comm is not None:
            self.world = comm.world
        else:
            self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if grad_reduce_dtype not in (None, torch.float32) and comm is None:
            raise ValueError("grad_reduce_dtype needs the RCCL communicator (comm=)")
        self.grad16 = (torch.empty(total, dtype=grad_reduce_dtype, device=dev)
                       if grad_reduce_dtype not in (None, torch.float32) else None)
        self.exchange_enabled = True        # False: skip the collectives (measures the step without its exchange)
        self.defer_exchange = False         # True: no bucket launches during backward; one exchange_all() after it

        # buckets: walk parameters in reverse registration order
        bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.bucket_ranges: List[List[int]] = []      # [lo, hi) element ranges
        bucket_of = [0] * len(params)
        hi = total
        for i in range(len(params) - 1, -1, -1):
            lo = self.offsets[i]
            bucket_of[i] = len(self.bucket_ranges)
            if hi - lo >= bucket_elems or i == 0:
                self.bucket_ranges.append([lo, hi])
                hi = lo
        self.bucket_size = [0] * len(self.bucket_ranges)
        self.sinks: List[GradSink] = []
        for i, p in enumerate(params):
            n, off = p.numel(), self.offsets[i]
            dview = self.data[off:off + n].view(p.shape)
            dview.copy_(p.data)                 # one-time setup copy
            p.data = dview
            gview = self.grad[off:off + n].view(p.shape)
            p.grad = gview
            s = GradSink(gview, bucket_of[i], self, i)
            p._dvt_sink = s
            if self.compute is not None:
                p._dvt_compute = self.compute[off:off + n].view(p.shape)
            self.sinks.append(s)
            self.bucket_size[bucket_of[i]] += 1
        self._pending = list(self.bucket_size)
        self._handles = []
        self._launched = [False] * len(self.bucket_ranges)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.step_count = 0
        # Parameters nobody wrote a gradient for are left alone by the optimizers, as torch leaves parameters whose
        # ``.grad`` is None (no weight decay, no moment update): one byte per 64-element block of the flat buffer
        # (every parameter slice is 64-element aligned), rebuilt only when the set of unwritten parameters changes.
        self.skip_mask = None
        self._skip_sig = ()
        self._packed = {}              # (parameter index, form, padded shape, dtype) -> (packed tensor, pack entry)
        self._packed_version = {}      # same key -> the parameter's _version when its packed form was last refreshed
        self._packed_valid = False

    # ------------------------------------------------------------------ compute copy
    def _after_step(self) -> None:
        from . import functional as F       # dropout generator: move past this step's sites (device-side add)
        F.next_step()
        self._packed_valid = False          # the packed convolution weights are stale from here on

    def sync_compute_copy(self) -> None:
        """One cast launch: fp32 masters -> bf16 copy used by the GEMMs."""
        if self.compute is not None:
            from . import _lib as L
            L.check(L.load().dvt_cast(self.data.data_ptr(), L.F32, self.compute.data_ptr(),
                                      ops._DT[self.compute_dtype], self.total, ops._stream()), "dvt_cast")
            self.compute_valid = True

    def invalidate_compute_copy(self) -> None:
        self.compute_valid = False
        self._packed_valid = False

    def invalidate_packed(self) -> None:
        """The packed convolution weights are stale (the next use repacks all of them in one launch)."""
        self._packed_valid = False

    def packed_weight(self, key, make, version: int = 0):
        """GEMM-operand forms of convolution weights (functional._packed_weight): kept here per (parameter, form) and ALL
        refreshed by ONE launch at their first use after the weights changed (an optimizer step, a broadcast, a loaded
        checkpoint) -- a training step then packs its convolution weights once instead of twice per layer.
        ``version``: the parameter's autograd version counter; an in-place edit the store did not make itself
        (``nn.Module.load_state_dict``, ``p.copy_`` / ``p.mul_`` under no_grad, a foreign torch optimizer) bumps it and
        invalidates every packed form.  Not seen: writes through ``p.data`` (a separate counter) and raw-pointer kernels
        -- call ``invalidate_compute_copy()`` after those.  A hipGraph that captures forward + backward only must begin
        with the cache invalid (``graph.capture_step`` does that) so that the repack launch is part of the graph."""
        e = self._packed.get(key)
        if e is None:
            dst, entry = make()
            self._packed[key] = (dst, entry)
            self._packed_version[key] = version
            ops.conv_weight_pack_group([entry])
            return dst
        if not self._packed_valid:
            # every form in one launch; each entry's source shares its parameter's version counter (detach()), so the
            # versions recorded here are the ones this launch has seen
            ops.conv_weight_pack_group([en for _, en in self._packed.values()])
            for k, (_, en) in self._packed.items():
                self._packed_version[k] = en[0]._version
            self._packed_valid = True
        if self._packed_version.get(key) != version:
            # an edit behind the store's back SINCE the last refresh: this entry alone (invalidating all of
            # them here made K stale forms cost K full-group launches)
            ops.conv_weight_pack_group([e[1]])
            self._packed_version[key] = version
        return e[0]

    # ------------------------------------------------------------------ per-step protocol
    def zero_grad(self) -> None:
        """Marks every sink fresh (first write overwrites): no memset pass."""
        self._pending = list(self.bucket_size)
        for s in self.sinks:
            s._fresh = True
            s._late_fresh = True
            s.late_written = False
            # A parameter nobody wrote in the previous step (a dead branch of the model: custom_resnet.py:149-153's avgpool + fc,
            # TPN.py:22's unused 1 x 1 convolution) will most likely not be written in this one either: its bucket does not
            # wait for it -- it waited until finish_backward, i.e. the whole bucket was exchanged AFTER backward (pyramid:
            # 21 MB at +0.02 ms).  The slice still holds the zeros of the last finish_backward.  Should it be written after
            # all: before the bucket's launch the write lands in place as usual, after it through the late-write redirect.
            # (Every rank runs the same model, so every rank takes the same decision: the collectives stay in one order.)
            s._uncounted = s.unwritten and s._zeroed
            if s._uncounted:
                self._pending[s.bucket] -= 1
        self._launched = [False] * len(self.bucket_ranges)
        self._handles = []
        for b, n in enumerate(self._pending):      # (a bucket of dead parameters only: exchanged at finish_backward as before)
            if n <= 0:
                self._pending[b] = 1 << 30

    def _launch_bucket(self, b: int) -> None:
        if self._launched[b] or self.defer_exchange:
            return
        self._launched[b] = True
        if self._exchanging():
            lo, hi = self.bucket_ranges[b]
            self._handles.append(self._all_reduce(self.grad[lo:hi], None if self.grad16 is None else self.grad16[lo:hi]))

    def _exchanging(self) -> bool:
        """With a communicator the collective also runs at world 1 (RCCL then copies in place): the single-rank rehearsal
        of the multi-GPU step executes the same enqueue / overlap / wait sequence as N ranks do."""
        return self.exchange_enabled and (self.world > 1 or self.comm is not None)

    def _all_reduce(self, t: torch.Tensor, via: Optional[torch.Tensor] = None):
        if self.comm is not None:
            return self.comm.all_reduce_async(t, via)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_first_write(self, sink: GradSink) -> None:
        b = sink.bucket
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch_bucket(b)

    def exchange_all(self) -> None:
        """Deferred form (``defer_exchange``): one all-reduce over the whole flat gradient, launched after backward."""
        if self._exchanging():
            self._all_reduce(self.grad, self.grad16).wait()

    def finish_backward(self, exchange: bool = True) -> None:
        """Call after loss.backward(): zero gradients nobody wrote (and exclude them from the optimizer step), flush
        remaining buckets, wait for the collectives (on the compute stream, not the host), fold in late writes.
        ``exchange=False`` (with ``defer_exchange``): local part only; ``exchange_all()`` follows."""
        from . import functional as F       # dgamma / dbeta reduces still deferred: one launch, then their sinks are written
        F.ln_flush(end_of_step=True)
        if self.comm is not None and self.comm.trace is not None:      # end of backward on the compute stream
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self.comm.marks["backward_end"] = ev
        unwritten = []
        for s in self.sinks:
            if s._fresh:
                if not s._zeroed:            # (still zero from the last step otherwise -- and its bucket may be in flight)
                    s._buf.zero_()           # memset of a slice nobody wrote this step (rare)
                    s._zeroed = True
                s._fresh = False
                s.unwritten = True
                unwritten.append(s.index)
        self._set_skip(tuple(unwritten))
        if self.defer_exchange:
            if exchange:
                self.exchange_all()
            return
        late = [s for s in self.sinks if s.late_written]
        for b in range(len(self.bucket_ranges)):
            self._launch_bucket(b)
        late_handles = [self._all_reduce(s._late) for s in late]
        for h in self._handles + late_handles:
            h.wait()
        for s in late:                       # reduced late contribution joins the reduced bucket
            s._buf.add_(s._late)
        self._handles = []

    def _set_skip(self, sig) -> None:
        if sig == self._skip_sig:
            return
        if self.data.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the set of parameters without a gradient changed inside a hipGraph capture; run the "
                               "warm-up steps with the same model mode as the captured step")
        self._skip_sig = sig
        if not sig:
            self.skip_mask = None
            return
        m = torch.zeros(self.total // _ALIGN, dtype=torch.uint8)
        for i in sig:
            lo = self.offsets[i] // _ALIGN
            hi = (self.offsets[i] + self.params[i].numel() + _ALIGN - 1) // _ALIGN
            m[lo:hi] = 1
        self.skip_mask = m.to(self.data.device)

    @property
    def loss_scale(self) -> float:
        """Multiply the loss (or its gradient) by this so that the summed gradients
        are the global-batch mean."""
        return 1.0 / self.world

    # ------------------------------------------------------------------ optimizer
    # ------------------------------------------------------------------ fp16 loss scaling (BASELINE configs[4])
    def enable_loss_scaling(self, init_scale: float = 65536.0, growth_interval: int = 2000, growth: float = 2.0,
                            backoff: float = 0.5) -> torch.Tensor:
        """Dynamic loss scaling with every piece of scaler state on the device (no host sync, hipGraph-capturable).
        Returns ``loss_grad``, the device scalar to seed backward with: ``loss.backward(flat.loss_grad)``; it holds
        ``scale / world`` and is refreshed by every ``adamw_step``.  A step whose all-reduced gradient contains a
        non-finite value is skipped on the device and halves the scale."""
        dev = self.data.device
        self.scaler = dict(growth_interval=int(growth_interval), growth=float(growth), backoff=float(backoff))
        self.scale_dev = torch.full((1,), float(init_scale), dtype=torch.float32, device=dev)
        self.found_inf = torch.zeros(1, dtype=torch.int32, device=dev)
        self.good_steps = torch.zeros(1, dtype=torch.int32, device=dev)
        self.loss_grad = torch.full((), float(init_scale) * self.loss_scale, dtype=torch.float32, device=dev)
        return self.loss_grad

    def adamw_step(self, lr: float, weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        """torch.optim.AdamW semantics (frame_transformer.py:127-129) in one launch.

        One deviation, by construction of the flat buffers: the bias-correction step count is GLOBAL (one device counter for
        all parameters), where torch keeps one per parameter and advances it only in steps in which that parameter has a
        gradient.  The two agree whenever every parameter is written in every step -- the reference's training loops (all
        modes of `FrameTransformer.training_step`, ViViT) -- and for parameters that never receive a gradient (skipped through
        `skip64`, state untouched, as torch skips `.grad is None`).  A parameter that receives gradients only in SOME steps would
        see the global count in its bias correction (1 - beta^t with t too large: a slightly larger effective step early on);
        `tests/test_gpu_dp.py` covers the always-written and never-written cases, not the intermittent one."""
        if self.exp_avg is None:
            self.init_optimizer_state()
        self.step_count += 1
        if getattr(self, "scaler", None) is not None:
            ops.adamw_step_scaled_(self.data, self.grad, self.exp_avg, self.exp_avg_sq, self.step_dev, self.scale_dev,
                                   self.found_inf, self.good_steps, self.loss_grad, lr=lr, beta1=betas[0],
                                   beta2=betas[1], eps=eps, weight_decay=weight_decay, loss_grad_base=self.loss_scale,
                                   skip=self.skip_mask, **self.scaler)
            self.sync_compute_copy()
            self._after_step()
            return
        # one launch: the update, the 16-bit mirror of the new weights, the step counter (on the device, so that the launch
        # can be captured in a hipGraph)
        ops.adamw_step_fused_(self.data, self.grad, self.exp_avg, self.exp_avg_sq, self.step_dev, lr=lr,
                              beta1=betas[0], beta2=betas[1], eps=eps, weight_decay=weight_decay, skip=self.skip_mask,
                              mirror=self.compute)
        if self.compute is not None:
            self.compute_valid = True
        self._packed_valid = False
        self._after_step()

    def sgd_step(self, lr: float, momentum: float = 0.0, weight_decay: float = 0.0) -> None:
        """torch.optim.SGD semantics (frame_transformer.py:124-126) in one launch over the flat buffers."""
        if momentum != 0.0 and getattr(self, "momentum_buf", None) is None:
            self.momentum_buf = torch.zeros_like(self.data)
        self.step_count += 1
        ops.sgd_step_(self.data, self.grad, getattr(self, "momentum_buf", None), lr=lr, momentum=momentum,
                      weight_decay=weight_decay, skip=self.skip_mask)
        self.sync_compute_copy()
        self._after_step()

    def adagrad_step(self, lr: float, weight_decay: float = 0.0, lr_decay: float = 0.0, eps: float = 1e-10) -> None:
        """torch.optim.Adagrad semantics (frame_transformer.py:130-132) in one launch over the flat buffers (the step count
        that enters `lr / (1 + (t - 1) lr_decay)` is global, as in `adamw_step`; the reference uses lr_decay = 0)."""
        if getattr(self, "state_sum", None) is None:
            self.state_sum = torch.zeros_like(self.data)
        self.step_count += 1
        ops.adagrad_step_(self.data, self.grad, self.state_sum, lr=lr, lr_decay=lr_decay, eps=eps,
                          weight_decay=weight_decay, step=self.step_count, skip=self.skip_mask)
        self.sync_compute_copy()
        self._after_step()

    def init_optimizer_state(self) -> None:
        self.exp_avg = torch.zeros_like(self.data)
        self.exp_avg_sq = torch.zeros_like(self.data)
        self.step_dev = torch.zeros(2, dtype=torch.int64, device=self.data.device)    # {steps taken, ticket of the fused step}

    # ------------------------------------------------------------------ checkpoint / resume of the optimizer side
    _STATE_TENSORS = ("exp_avg", "exp_avg_sq", "step_dev", "momentum_buf", "state_sum", "scale_dev", "good_steps")

    def state_dict(self) -> dict:
        """Optimizer-side state of the flat buffers (the parameters themselves are saved through the module's own
        ``state_dict``, whose tensors are views of ``self.data``).  Tensors are cloned to the host."""
        sd = {"step_count": self.step_count, "total": self.total}
        for k in self._STATE_TENSORS:
            t = getattr(self, k, None)
            if t is not None:
                sd[k] = t.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd: dict) -> None:
        if sd.get("total") != self.total:
            raise ValueError(f"flat-buffer layout mismatch: checkpoint has {sd.get('total')} elements, model has {self.total}")
        self.step_count = int(sd["step_count"])
        dev = self.data.device
        if "scale_dev" in sd and getattr(self, "scaler", None) is None:
            self.enable_loss_scaling(float(sd["scale_dev"]))
        for k in self._STATE_TENSORS:
            if k in sd:
                cur = getattr(self, k, None)
                src = sd[k].to(dev)
                if k == "step_dev":                      # {steps, ticket}; older checkpoints hold the count alone
                    src = torch.stack((src.reshape(-1)[0], torch.zeros((), dtype=torch.int64, device=dev)))
                if cur is None:
                    setattr(self, k, src.clone())
                else:
                    cur.copy_(src)
        if getattr(self, "scaler", None) is not None:
            self.loss_grad.copy_(self.scale_dev.reshape(()) * self.loss_scale)
        self.invalidate_compute_copy()

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.comm is not None:
            self.comm.broadcast(self.data, src)
        elif self.world > 1:
            dist.broadcast(self.data, src=src, group=self.group)
        self.invalidate_compute_copy()
