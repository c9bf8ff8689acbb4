"""Feasibility probe: one optimizer step of the ViViT metric shape as TWO micro-batches on two streams, staggered so that
one micro-batch's small-launch zone (last space layer on the CLS rows, temporal encoder, heads, loss and their backward:
~130 launches of 4-12 us) runs beside the other's full-size kernels.  Compares gradients and graph-replay time with the
single-stream step.   python3 tools/dev/pipeline_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dvt_amd  # noqa: E402
from dvt_amd import functional as F  # noqa: E402
from dvt_amd.dp import FlatParameters  # noqa: E402
from dvt_amd.graph import capture_step  # noqa: E402
from dvt_amd.models.vit import ViViT  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, T, d = 8, 32, 512
torch.manual_seed(1130)
net = ViViT(224, 16, 19, T, dim=d, depth=4, heads=8, dim_head=64, compute_dtype=torch.bfloat16).cuda().train()
flat = FlatParameters(net, compute_dtype=torch.bfloat16)
flat.sync_compute_copy()
gen = torch.Generator().manual_seed(7)
x = torch.randn(B, T, 3, 224, 224, generator=gen).bfloat16().cuda()
y = (torch.rand(B, 19, generator=gen) < 0.2).float().cuda()
g1 = torch.full((), 1.0, device="cuda")
gh = torch.full((), 0.5, device="cuda")


def fwd_big(xp):
    b, t = xp.shape[0], xp.shape[1]
    n = (xp.shape[3] // net.patch_size) * (xp.shape[4] // net.patch_size)
    pe = net.to_patch_embedding[1]
    emb = F.patch_embed(xp, pe.weight, pe.bias, net.patch_size, net.compute_dtype)
    u = F.tokens_assemble(emb, net.space_token, net.pos_embedding, b * t, t, n)
    for a, f in list(net.space_transformer.layers)[:-1]:
        u = a.fn(u, _norm=a.norm, _residual=True)
        u = f.fn(u, _norm=f.norm, _residual=True)
    return u


def fwd_small(u, b, t):
    st, tt = net.space_transformer, net.temporal_transformer
    attn, ff = st.layers[-1]
    an, af = attn.norm, attn.fn
    c = F.attn_block_cls(u, an.weight, an.bias, af.to_qkv.weight, af.to_out[0].weight, af.to_out[0].bias, af.heads, eps=an.eps)
    s = ff.fn(c, _norm=ff.norm, _residual=True).view(b * t, 1, -1)
    sn = st.norm
    seq = F.cls_norm_concat(s, sn.weight, sn.bias, net.temporal_token, b, t, sn.eps)
    pooled = F.layernorm(tt.forward_layers_cls(seq), tt.norm.weight, tt.norm.bias, tt.norm.eps)
    hn, hl = net.mlp_head[0], net.mlp_head[1]
    return F.linear(F.layernorm(pooled, hn.weight, hn.bias, hn.eps), hl.weight, hl.bias, out_f32=True)


def fwd_bwd_single():
    flat.zero_grad()
    loss = F.bce_with_logits(net(x), y)
    loss.backward(g1)
    flat.finish_backward()
    return loss


side = torch.cuda.Stream()


def fwd_bwd_pipe():
    flat.zero_grad()
    main = torch.cuda.current_stream()
    h = B // 2
    xa, ya, xb, yb = x[:h], y[:h], x[h:], y[h:]
    ua = fwd_big(xa)
    e_fwd = torch.cuda.Event()
    e_fwd.record(main)
    side.wait_event(e_fwd)
    with torch.cuda.stream(side):
        ub = fwd_big(xb)
    ua_d = ua.detach().requires_grad_(True)
    la = F.bce_with_logits(fwd_small(ua_d, h, T), ya)
    la.backward(gh)
    e_small = torch.cuda.Event()
    e_small.record(main)
    with torch.cuda.stream(side):
        side.wait_event(e_small)
        ub_d = ub.detach().requires_grad_(True)
        lb = F.bce_with_logits(fwd_small(ub_d, h, T), yb)
        lb.backward(gh)
    ua.backward(ua_d.grad)
    e_done = torch.cuda.Event()
    e_done.record(main)
    with torch.cuda.stream(side):
        side.wait_event(e_done)
        ub.backward(ub_d.grad)
    main.wait_stream(side)
    flat.finish_backward()
    return la


def step_single():
    loss = fwd_bwd_single()
    flat.adamw_step(lr=5e-6, weight_decay=0.09)
    return loss


def step_pipe():
    loss = fwd_bwd_pipe()
    flat.adamw_step(lr=5e-6, weight_decay=0.09)
    return loss


# gradients: pipelined vs single
fwd_bwd_single()
torch.cuda.synchronize()
ref = flat.grad.clone()
fwd_bwd_pipe()
torch.cuda.synchronize()
got = flat.grad.clone()
print("gradient rel-L2 pipe vs single:", float((got - ref).norm() / ref.norm()), "max abs", float((got - ref).abs().max()))


def timed(name, fn):
    replay, _ = capture_step(fn, warmup=2)
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / it:7.3f} ms/step")


timed("single", step_single)
timed("pipelined", step_pipe)
timed("single", step_single)
timed("pipelined", step_pipe)


def fwd_bwd_seq():      # the two micro-batches one after the other on ONE stream (no overlap): the cost of half-size kernels
    flat.zero_grad()
    h = B // 2
    for xs, ys in ((x[:h], y[:h]), (x[h:], y[h:])):
        F.bce_with_logits(net(xs), ys).backward(gh)
    flat.finish_backward()


def step_seq():
    fwd_bwd_seq()
    flat.adamw_step(lr=5e-6, weight_decay=0.09)
    return g1


timed("two halves", step_seq)

# do two branches of a captured graph run concurrently at all?
from dvt_amd import ops  # noqa: E402


def two_delays():
    main = torch.cuda.current_stream()
    e = torch.cuda.Event()
    e.record(main)
    side.wait_event(e)
    ops.device_delay(300)
    with torch.cuda.stream(side):
        ops.device_delay(300)
    main.wait_stream(side)
    return g1


def one_delay():
    ops.device_delay(300)
    return g1


timed("1 x 300us", one_delay)
timed("2 x 300us", two_delays)
