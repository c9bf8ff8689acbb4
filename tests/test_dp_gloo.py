"""Data-parallel gradient exchange on CPU (gloo, world_size 2).

The GPU kernels cannot run here, so each rank computes the gradients of its half of
the batch with the CPU oracle and hands them to ``dp.FlatParameters`` through the
same GradSink protocol the HIP backward uses (first write overwrites, buckets fire
when complete, sum all-reduce, 1/world folded into the loss).  The all-reduced flat
gradient must equal the single-process full-batch gradient.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import clip_path as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    from dvt_amd.models.vit import ViViT
    torch.manual_seed(1130)
    return ViViT(32, 8, 19, 3, dim=64, depth=1, heads=2, dim_head=32, compute_dtype=torch.float32)


def _data():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 3, 32, 32, generator=g)
    y = (torch.rand(4, 19, generator=g) < 0.3).float()
    return x, y


def _oracle_grads(net, x, y, scale):
    P = {k: v.detach().clone().requires_grad_(True) for k, v in net.named_parameters()}
    logits = O.vivit_forward(x, P, patch=8, depth=1, heads=2)
    (O.bce_with_logits(logits, y) * scale).backward()
    return {k: v.grad for k, v in P.items()}


def _worker(rank, world, port, bucket_mb, out, twice=()):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvt_amd.dp import FlatParameters
    net = _model()
    if rank == 1:                      # ranks start from different weights: broadcast must fix it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    flat = FlatParameters(net, bucket_mb=bucket_mb, compute_dtype=None)
    flat.broadcast_parameters(0)
    x, y = _data()
    xs, ys = x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]
    for step in range(2):              # second step: stale gradients must be overwritten
        flat.zero_grad()
        grads = _oracle_grads(net, xs, ys, flat.loss_scale)
        names = [k for k, _ in net.named_parameters()]
        def write(k, part):
            s = dict(net.named_parameters())[k]._dvt_sink
            if s.fresh:                # the writer protocol of functional.py: buf / fresh, then mark_written
                s.buf.copy_(part)
            else:
                s.buf.add_(part)
            s.mark_written()

        for k in reversed(names):      # backward order
            if k == "temporal_token" and step == 0:
                continue               # a parameter nobody writes: finish_backward zero-fills it
            write(k, grads[k] * (0.25 if k in twice else 1.0))
        for k in twice:                # a parameter used twice: its second contribution arrives when its bucket's
            write(k, grads[k] * 0.75)  # all-reduce is long in flight (tiny buckets) -> the late-write path
        flat.finish_backward()
        if step == 0:
            i = names.index("temporal_token")
            assert flat.sinks[i].unwritten and flat.skip_mask is not None
            lo = flat.offsets[i] // 64
            assert int(flat.skip_mask.sum()) == (flat.params[i].numel() + 63) // 64 and int(flat.skip_mask[lo]) == 1
        else:
            assert flat.skip_mask is None and not any(s.unwritten for s in flat.sinks)
        if twice and bucket_mb < 1:
            assert all(dict(net.named_parameters())[k]._dvt_sink.late_written for k in twice)
    if rank == 0:
        torch.save({"grad": flat.grad.clone(), "data": flat.data.clone(), "nb": len(flat.bucket_ranges),
                    "names": names, "offsets": flat.offsets}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb,twice", [(32.0, ()), (0.05, ()),
                                             (0.05, ("mlp_head.1.weight", "space_transformer.layers.0.1.fn.net.0.bias")),
                                             (32.0, ("mlp_head.1.weight",))])
def test_two_rank_allreduce_equals_full_batch(tmp_path, bucket_mb, twice):
    """twice: parameters written a second time AFTER every other gradient (a shared head / an encoder called on two
    streams).  With small buckets the second write finds its bucket already launched; it must neither be added on top
    of the reduced sum nor race with the collective (dp.py) -- the result is still the full-batch gradient."""
    import dvt_amd  # noqa: F401  (registers the package alias before spawn pickles the worker)
    port = _free_port()
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, bucket_mb, out, twice), nprocs=2, join=True)
    res = torch.load(out)
    net = _model()
    x, y = _data()
    ref = _oracle_grads(net, x, y, 1.0)                 # single process, full batch of 4
    params = dict(net.named_parameters())
    assert res["nb"] >= (1 if bucket_mb > 1 else 3)
    for k, off in zip(res["names"], res["offsets"]):
        n = params[k].numel()
        got = res["grad"][off:off + n].view(params[k].shape)
        assert torch.allclose(got, ref[k], rtol=1e-4, atol=1e-6), k
        # broadcast made rank 1's (perturbed) weights equal rank 0's
        assert torch.equal(res["data"][off:off + n].view(params[k].shape), params[k].detach())


BENCH_BUCKET_MB = 13.0        # bench.py --bucket-mb default: the plan the first 8-GPU run will use


def _metric_model():
    from dvt_amd.models.vit import ViViT
    torch.manual_seed(1130)
    return ViViT(224, 16, 19, 32, dim=512, depth=4, heads=8, dim_head=64, compute_dtype=torch.float32)


def _expected_plan(sizes, bucket_mb):
    """Independent restatement of the bucket rule: walk the parameters in REVERSE registration order
    (= backward completion order), close a bucket once it holds >= bucket_mb of fp32 gradients; 64-element aligned slices."""
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    limit, ranges, hi = int(bucket_mb * (1 << 20) / 4), [], total
    for i in range(len(sizes) - 1, -1, -1):
        if hi - offs[i] >= limit or i == 0:
            ranges.append([offs[i], hi])
            hi = offs[i]
    return ranges, total


def _plan_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvt_amd.dp import FlatParameters
    net = _metric_model()
    flat = FlatParameters(net, bucket_mb=BENCH_BUCKET_MB, compute_dtype=None)
    named = list(net.named_parameters())
    twice = "space_transformer.norm.weight"          # written again after its bucket has been launched (late buffer)
    never = "temporal_token"                         # nobody writes it in step 0 (zero fill + skip mask)
    launched_at = {}
    i_never = [k for k, _ in named].index(never)
    for step in range(3):                            # steps 0, 1: `never` unwritten; step 2: written after all
        flat.zero_grad()
        for i in range(len(named) - 1, -1, -1):      # backward completion order
            k, p = named[i]
            if k == never and step < 2:
                continue
            s = p._dvt_sink
            val = (rank + 1) * (i + 1) * 1e-3 * flat.loss_scale * (0.25 if k == twice else 1.0)
            assert s.fresh
            s.buf.fill_(val)
            s.mark_written()
            if step == 0:
                launched_at[i] = sum(flat._launched)
        # the bucket of a never-written parameter: in step 0 it waits for finish_backward; from step 1 on nobody expects the
        # parameter any more and the bucket leaves with its last WRITTEN parameter (its slice holds step 0's zeros)
        assert flat._launched[flat.sinks[i_never].bucket] == (step > 0), step
        s = dict(named)[twice]._dvt_sink               # the second write: its bucket's all-reduce is in flight
        i2 = [k for k, _ in named].index(twice)
        assert flat._launched[s.bucket] and s.fresh    # (fresh now names the late buffer)
        s.buf.fill_((rank + 1) * (i2 + 1) * 1e-3 * flat.loss_scale * 0.75)
        s.mark_written()
        assert s.late_written
        flat.finish_backward()
        if step < 2:
            i = i_never
            assert flat.sinks[i].unwritten and flat.skip_mask is not None and int(flat.skip_mask.sum()) == (named[i][1].numel() + 63) // 64
            assert float(named[i][1].grad.abs().max()) == 0.0
        else:                 # written in step 2 after all: in place (its bucket had not left yet) or through the late path
            assert flat.skip_mask is None and not flat.sinks[i_never].unwritten
    if rank == 0:
        torch.save({"ranges": flat.bucket_ranges, "offsets": flat.offsets, "total": flat.total,
                    "launched_at": launched_at, "bucket_size": flat.bucket_size,
                    "grad_means": [float(p.grad.double().mean()) for _, p in named],
                    "grad_spread": [float((p.grad.max() - p.grad.min()).abs()) for _, p in named]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_bucket_plan_at_the_metric_parameter_set(tmp_path):
    """The exact gradient-exchange plan of the headline run -- bench.py's 13 MiB buckets over the d = 512 ViViT's 28.8 M
    parameters -- at world 2 over gloo: bucket boundaries against an independent restatement of the rule, buckets fire in
    backward completion order as their last parameter is written, a twice-written parameter goes through the late buffer,
    a never-written one through the zero fill and the skip mask, and the reduced flat gradient is the rank mean."""
    import dvt_amd  # noqa: F401
    port = _free_port()
    out = str(tmp_path / "plan.pt")
    mp.spawn(_plan_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    net = _metric_model()
    sizes = [p.numel() for p in net.parameters()]
    want, total = _expected_plan(sizes, BENCH_BUCKET_MB)
    assert res["total"] == total and [list(r) for r in res["ranges"]] == want
    mb = [(hi - lo) * 4 / (1 << 20) for lo, hi in want]
    print(f"[bucket plan] {len(want)} buckets (MiB, in launch order): " + ", ".join(f"{m:.1f}" for m in mb))
    assert len(want) == 7 and all(m >= BENCH_BUCKET_MB for m in mb[:-1]) and sum(res["bucket_size"]) == len(sizes)
    # a bucket is launched exactly when its lowest-offset parameter (the last one written in backward order) is written;
    # the bucket holding the never-written parameter waits for finish_backward
    names = [k for k, _ in net.named_parameters()]
    i_never = names.index("temporal_token")
    starts = {lo for lo, hi in want if not (lo <= res["offsets"][i_never] < hi)}
    fired = 0
    for i in range(len(sizes) - 1, -1, -1):
        if i == i_never:
            continue
        if res["offsets"][i] in starts:
            fired += 1
        assert res["launched_at"][i] == fired, i
    assert fired == len(want) - 1
    for i, (m, sp) in enumerate(zip(res["grad_means"], res["grad_spread"])):
        assert sp == 0.0 and abs(m - 1.5 * (i + 1) * 1e-3) < 1e-6 * (i + 1), i      # mean over ranks of (rank + 1) * (i + 1) * 1e-3


def test_bucket_layout_covers_every_parameter_once():
    from dvt_amd.dp import FlatParameters
    net = _model()
    flat = FlatParameters(net, bucket_mb=0.02, compute_dtype=None)
    covered = np.zeros(flat.total, dtype=np.int32)
    for lo, hi in flat.bucket_ranges:
        covered[lo:hi] += 1
    assert (covered == 1).all()
    assert sum(flat.bucket_size) == len(flat.params)
    # parameters are views of the flat buffers
    p0 = flat.params[0]
    assert p0.data.data_ptr() == flat.data.data_ptr() and p0.grad.data_ptr() == flat.grad.data_ptr()
    # buckets are formed from the END of the buffer (backward completion order)
    assert flat.bucket_ranges[0][1] == flat.total and flat.bucket_ranges[-1][0] == 0


def _gather_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvt_amd.metrics import gather_rows
    rows = 3 if rank == 0 else 5                        # ranks hold different numbers of validation samples
    t = torch.arange(rows * 4, dtype=torch.float32).view(rows, 4) + 100 * rank
    g = gather_rows(t)
    if rank == 0:
        out.put(g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_eval_accumulators_gather_across_ranks():
    """metrics.gather_rows: every rank ends up with all ranks' running_logits rows, in rank order."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    b = np.arange(20, dtype=np.float32).reshape(5, 4) + 100
    assert np.array_equal(got, np.concatenate([a, b]))


def test_flat_parameters_checkpoint_roundtrip():
    """Resume: module state_dict (views of the flat buffer) + FlatParameters.state_dict (moments, step) restore a
    second instance exactly (CPU plumbing; the kernels are not involved)."""
    from dvt_amd.dp import FlatParameters
    a = _model()
    fa = FlatParameters(a, compute_dtype=None)
    fa.init_optimizer_state()
    fa.exp_avg.uniform_(-1, 1); fa.exp_avg_sq.uniform_(0, 1); fa.step_dev[0] = 7; fa.step_count = 7
    with torch.no_grad():
        fa.data.add_(0.25)
    b = _model()
    fb = FlatParameters(b, compute_dtype=None)
    b.load_state_dict(a.state_dict())
    fb.load_state_dict(fa.state_dict())
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):      # (alignment gaps of the flat buffer are not state)
        assert torch.equal(p, q), k
    assert torch.equal(fb.exp_avg, fa.exp_avg) and torch.equal(fb.exp_avg_sq, fa.exp_avg_sq)
    assert int(fb.step_dev[0]) == 7 and int(fb.step_dev[1]) == 0 and fb.step_count == 7
    bad = fa.state_dict(); bad["total"] = 1
    with pytest.raises(ValueError, match="layout"):
        fb.load_state_dict(bad)
