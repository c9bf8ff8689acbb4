"""Data parallelism on the real kernels: two ranks share the one GPU of the test box (gloo carries the gradient
all-reduce, as NCCL/RCCL refuses two ranks on one device); each rank runs the HIP forward/backward on its half of the
batch and the all-reduced flat gradient must equal the single-process full-batch HIP gradient (fp32 kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _model():
    from dvt_amd.models.vit import ViViT
    torch.manual_seed(1130)
    return ViViT(32, 8, 19, 3, dim=64, depth=2, heads=2, dim_head=32, compute_dtype=torch.float32)


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(4, 3, 3, 32, 32, generator=g), (torch.rand(4, 19, generator=g) < 0.3).float()


def _step(net, flat, x, y):
    from dvt_amd import functional as F
    flat.zero_grad()
    loss = net.loss(x.cuda(), y.cuda())[0]
    loss.backward(torch.full((), flat.loss_scale, device="cuda"))
    flat.finish_backward()
    return loss


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvt_amd.dp import FlatParameters
    net = _model().cuda()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.5)                              # broadcast must repair this
    flat = FlatParameters(net, bucket_mb=0.1, compute_dtype=None)
    flat.broadcast_parameters(0)
    x, y = _data()
    _step(net, flat, x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2])
    flat.adamw_step(lr=1e-3, weight_decay=0.09)
    if rank == 0:
        torch.save({"grad": flat.grad.cpu(), "data": flat.data.cpu(), "nb": len(flat.bucket_ranges)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_hip_kernels_equal_full_batch(device, tmp_path):
    import dvt_amd  # noqa: F401
    from dvt_amd.dp import FlatParameters
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["nb"] >= 2                                # several buckets: the overlapped launch path ran
    net = _model().cuda()
    flat = FlatParameters(net, compute_dtype=None)
    x, y = _data()
    _step(net, flat, x, y)                               # single process, full batch
    ref_grad = flat.grad.cpu()
    flat.adamw_step(lr=1e-3, weight_decay=0.09)
    scale = ref_grad.abs().max()
    assert float((res["grad"] - ref_grad).abs().max() / scale) < 2e-5
    assert float((res["data"] - flat.data.cpu()).abs().max()) < 1e-5


@pytest.mark.parametrize("opt", ["adamw", "sgd", "adagrad", "adamw_scaled"])
def test_parameter_without_gradient_is_left_alone_by_the_flat_optimizers(device, opt):
    """torch's optimizers (and the reference's configure_optimizers, frame_transformer.py:123-134) skip parameters
    whose .grad is None.  FlatParameters gives every parameter a permanent gradient view and zero-fills the ones
    nobody wrote, so the fused flat-buffer steps must exclude them explicitly: with weight decay 0.09 a frozen encoder
    would otherwise drift towards zero.  Every ViViT parameter is used by its forward, so a registered but
    unused probe parameter plays the frozen encoder."""
    from dvt_amd.dp import FlatParameters
    from dvt_amd import functional as F
    net = _model().cuda()
    extra = torch.nn.Parameter(torch.randn(300, device="cuda"))           # registered, never used by forward
    net.register_parameter("unused_probe", extra)
    flat = FlatParameters(net, compute_dtype=None)
    if opt == "adamw_scaled":
        seed = flat.enable_loss_scaling(init_scale=8.0, growth_interval=1000)
    else:
        seed = torch.ones((), device="cuda")
    x, y = _data()
    before = net.unused_probe.detach().clone()
    used_before = net.mlp_head[1].weight.detach().clone()
    for _ in range(3):
        flat.zero_grad()
        net.loss(x.cuda(), y.cuda())[0].backward(seed)
        flat.finish_backward()
        if opt in ("adamw", "adamw_scaled"):
            flat.adamw_step(lr=1e-2, weight_decay=0.09)
        elif opt == "sgd":
            flat.sgd_step(lr=1e-2, momentum=0.9, weight_decay=0.09)
        else:
            flat.adagrad_step(lr=1e-2, weight_decay=0.09)
    assert flat.skip_mask is not None and int(flat.skip_mask.sum()) == (300 + 63) // 64
    assert torch.equal(net.unused_probe.detach(), before)                 # bit-unchanged: no decay, no update
    assert not torch.equal(net.mlp_head[1].weight.detach(), used_before)  # the others did move
    i = [k for k, _ in net.named_parameters()].index("unused_probe")
    lo, n = flat.offsets[i], 300
    for name in ("exp_avg", "exp_avg_sq", "momentum_buf", "state_sum"):
        st = getattr(flat, name, None)
        if st is not None:
            assert float(st[lo:lo + n].abs().max()) == 0.0, name
    # the torch.optim-interface optimizers skip it as well (p.grad is a permanent view, never None)
    from dvt_amd import optim
    o = optim.AdamW(net.parameters(), lr=1e-2, weight_decay=0.09)
    flat.zero_grad()
    net.loss(x.cuda(), y.cuda())[0].backward(seed)
    flat.finish_backward()
    o.step()
    assert torch.equal(net.unused_probe.detach(), before) and net.unused_probe not in o.state


def _make_step(net, flat, x, y):
    from dvt_amd import functional as F
    seed = torch.full((), flat.loss_scale, device="cuda")

    def fwd_bwd():
        flat.zero_grad()
        loss = net.loss(x, y)[0]
        loss.backward(seed)
        return loss

    def update():
        flat.adamw_step(lr=1e-3, weight_decay=0.09)

    def step():
        loss = fwd_bwd()
        flat.finish_backward()
        update()
        return loss

    return fwd_bwd, update, step


@pytest.mark.parametrize("grad_dtype", [None, torch.bfloat16])
def test_rccl_communicator_behind_the_c_abi_and_graph_captured_dp_step(device, grad_dtype):
    """dvt_comm_* (RCCL behind the C ABI) on the one GPU of the test box: a single-rank communicator runs the same
    enqueue / side-stream / event-join sequence as N ranks do.  (1) all-reduce and broadcast leave the data intact at
    world 1, in fp32 and through the half-width wire format; (2) the data-parallel step with its bucketed all-reduces
    captured INSIDE the hipGraph replays to the eager step's weights bit for bit; (3) so does the two-graph form around
    an eagerly launched exchange."""
    import os
    import torch.distributed as dist
    from dvt_amd.dp import Communicator, FlatParameters
    from dvt_amd.graph import capture_step, capture_step_segments
    comm = Communicator(1, 0, Communicator.unique_id())
    try:
        t = torch.randn(100003, device="cuda")
        ref = t.clone()
        comm.all_reduce_async(t).wait()
        comm.broadcast(t, 0)
        torch.cuda.synchronize()
        assert torch.equal(t, ref)
        half = torch.empty(100003, dtype=torch.bfloat16, device="cuda")
        comm.all_reduce_async(t, half).wait()
        torch.cuda.synchronize()
        assert torch.equal(t, ref.bfloat16().float())              # the wire format is bf16: one rounding, nothing else
        x, y = _data()
        x, y = x.cuda(), y.cuda()
        results = {}
        for mode in ("eager", "graph", "segments"):
            net = _model().cuda()
            flat = FlatParameters(net, bucket_mb=0.1, compute_dtype=None, comm=comm, grad_reduce_dtype=grad_dtype)
            assert len(flat.bucket_ranges) >= 2
            fwd_bwd, update, step = _make_step(net, flat, x, y)
            losses = []
            if mode == "eager":
                for i in range(2 + 3):
                    l = step()
                    if i >= 2:
                        losses.append(float(l.detach()))
            elif mode == "graph":
                replay, out = capture_step(step, warmup=2)
                for _ in range(3):
                    replay()
                    losses.append(float(out.detach()))
            else:
                flat.defer_exchange = True

                def local():
                    l = fwd_bwd()
                    flat.finish_backward(exchange=False)
                    return l

                replay, out = capture_step_segments(local, flat.exchange_all, update, warmup=2)
                for _ in range(3):
                    replay()
                    losses.append(float(out.detach()))
            results[mode] = (losses, flat.data.clone(), flat.grad.clone())
        assert results["graph"][0] == results["eager"][0] and torch.equal(results["graph"][1], results["eager"][1])
        assert results["segments"][0] == results["eager"][0] and torch.equal(results["segments"][1], results["eager"][1])
        if grad_dtype is not None:                                   # the gradients really went through the half-width format
            assert torch.equal(results["eager"][2], results["eager"][2].bfloat16().float())
    finally:
        comm.destroy()


def _graph_worker(rank, world, port, out):
    """Two gloo ranks on one GPU: gloo cannot be captured, so the graph form is the two-graph one around the exchange."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvt_amd.dp import FlatParameters
    from dvt_amd.graph import capture_step_segments
    x, y = _data()
    xs, ys = x[rank * 2:(rank + 1) * 2].cuda(), y[rank * 2:(rank + 1) * 2].cuda()
    res = {}
    for mode in ("eager", "segments"):
        net = _model().cuda()
        flat = FlatParameters(net, bucket_mb=0.1, compute_dtype=None)
        flat.broadcast_parameters(0)
        fwd_bwd, update, step = _make_step(net, flat, xs, ys)
        if mode == "eager":
            for _ in range(2 + 3):
                step()
        else:
            flat.defer_exchange = True

            def local():
                l = fwd_bwd()
                flat.finish_backward(exchange=False)
                return l

            replay, _ = capture_step_segments(local, flat.exchange_all, update, warmup=2)
            for _ in range(3):
                replay()
        torch.cuda.synchronize()
        res[mode] = flat.data.cpu()
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_graph_segments_equal_eager_bit_for_bit(device, tmp_path):
    """the graph-replayed data-parallel step on 2 ranks == the eager one, bit for bit (deferred exchange:
    one all-reduce over the flat gradient between the fwd+bwd graph and the optimizer graph)."""
    import dvt_amd  # noqa: F401
    out = str(tmp_path / "g.pt")
    mp.spawn(_graph_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert torch.equal(res["eager"], res["segments"])


def test_twice_used_layernorm_with_gamma_and_beta_in_different_buckets(device):
    """a LayerNorm used twice in a step whose gamma and beta sit in DIFFERENT gradient buckets.  Registration
    order [W_pre, A, ln_weight, ln_bias, X] with 128-element buckets gives {ln.bias, X} and {A, ln.weight}; forward
    h = LN(linear(x, W_pre, X)); y = LN(linear(h, A)).  In backward the second use writes gamma / beta first, A's weight
    gradient then completes (and launches) gamma's bucket, and the first use's LayerNorm backward finds gamma redirected
    to its late buffer (accumulate = False) while beta still accumulates into the bucket (X is unwritten): the two sinks
    need their own accumulate flags.  Two steps (a stale late buffer would show in the second) against torch autograd."""
    from dvt_amd.dp import Communicator, FlatParameters
    from dvt_amd import functional as F
    from oracle import clip_path as O
    d = 64

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(3)
            self.w_pre = torch.nn.Parameter(torch.randn(d, d, generator=g) / 8)
            self.a = torch.nn.Parameter(torch.randn(d, d, generator=g) / 8)
            self.ln_weight = torch.nn.Parameter(1 + 0.1 * torch.randn(d, generator=g))     # (direct parameters: registration
            self.ln_bias = torch.nn.Parameter(0.1 * torch.randn(d, generator=g))          #  order = this order)
            self.x_bias = torch.nn.Parameter(0.1 * torch.randn(d, generator=g))

        def forward(self, x):
            h = F.layernorm(F.linear(x, self.w_pre, self.x_bias), self.ln_weight, self.ln_bias, 1e-5)
            return F.layernorm(F.linear(h, self.a), self.ln_weight, self.ln_bias, 1e-5)

    comm = Communicator(1, 0, Communicator.unique_id())
    try:
        net = Net().cuda()
        flat = FlatParameters(net, bucket_mb=128 * 4 / (1 << 20), compute_dtype=None, comm=comm)
        names = [n for n, _ in net.named_parameters()]
        bucket = {n: flat.sinks[i].bucket for i, n in enumerate(names)}
        assert (bucket["ln_weight"] != bucket["ln_bias"] and bucket["ln_bias"] == bucket["x_bias"]
                and bucket["ln_weight"] == bucket["a"]), bucket
        gen = torch.Generator().manual_seed(4)
        for step in range(2):
            x = torch.randn(16, d, generator=gen)
            t = torch.randn(16, d, generator=gen)
            P = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in net.named_parameters()}
            h = O.layernorm(O.linear(x, P["w_pre"], P["x_bias"]), P["ln_weight"], P["ln_bias"])
            ref = O.layernorm(O.linear(h, P["a"]), P["ln_weight"], P["ln_bias"])
            ref.backward(t)
            flat.zero_grad()
            net(x.cuda()).backward(t.cuda())
            flat.finish_backward()
            torch.cuda.synchronize()
            for n, p in net.named_parameters():
                e = float((p.grad.cpu() - P[n].grad).norm() / P[n].grad.norm())
                assert e < 1e-4, (step, n, e)
    finally:
        comm.destroy()
