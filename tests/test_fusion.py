"""CollaborativeGating (collabgating.py) and ContrastiveLoss / NT-Xent (losses/ntxent.py).
CPU: oracle vs the reference-generated fixture; gloo world_size-2 check of the gathered NT-Xent.  GPU: HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fusion_path as FP
from tests.util import golden, rel_l2, check_grad_digest

T = lambda a: torch.from_numpy(np.asarray(a))


def _cg_inputs(g):
    return [[[T(g[f"cg_x:{b}:{s}:{e}"]) for e in range(3)] for s in range(2)] for b in range(2)]


def _cg_weights(g):
    rng = np.random.default_rng(int(g["cg_wseed"]))
    shapes = {"projection.weight": (2048, 2048), "projection.bias": (2048,), "geu.fc.weight": (1024, 2048),
              "geu.fc.bias": (1024,)}
    P = {}
    for n in [str(x) for x in g["cg_wnames"]]:
        a = rng.standard_normal(shapes[n]).astype(np.float32)
        P[n] = torch.from_numpy(a * np.float32(0.02 if len(shapes[n]) == 2 else 0.1))
    return P


def test_oracle_contrastive_matches_reference():
    g = golden("fusion.npz")
    zi, zj = T(g["cl_zi"]).requires_grad_(True), T(g["cl_zj"]).requires_grad_(True)
    loss = FP.contrastive_loss(zi, zj, 0.5)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["cl_loss"][0])) < 1e-6
    assert rel_l2(zi.grad, T(g["cl_gzi"])) < 1e-5 and rel_l2(zj.grad, T(g["cl_gzj"])) < 1e-5


def test_oracle_collaborative_gating_matches_reference():
    g = golden("fusion.npz")
    P = {k: v.requires_grad_(True) for k, v in _cg_weights(g).items()}
    y = FP.collaborative_gating(_cg_inputs(g), P)
    assert rel_l2(y, T(g["cg_out"])) < 2e-6
    (y * T(g["cg_gy"])).sum().backward()
    check_grad_digest(g, {k: v.grad for k, v in P.items()}, 2e-4, "collabgating")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_hip_contrastive_loss(device, dtype, tol):
    from dvt_amd.models.losses.ntxent import ContrastiveLoss, NT_Xent
    g = golden("fusion.npz")
    zi = T(g["cl_zi"]).to(dtype).cuda().requires_grad_(True)
    zj = T(g["cl_zj"]).to(dtype).cuda().requires_grad_(True)
    loss = ContrastiveLoss(6, 0.5).cuda()(zi, zj)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["cl_loss"][0])) < 50 * tol
    assert rel_l2(zi.grad, T(g["cl_gzi"])) < 10 * tol and rel_l2(zj.grad, T(g["cl_gzj"])) < 10 * tol
    again = NT_Xent(6, 0.5, 1)(zi.detach(), zj.detach())          # same objective, returned (the reference returns None)
    assert abs(float(again) - float(loss.detach())) < 1e-6
    with pytest.raises(ValueError):
        ContrastiveLoss(5, 0.5).cuda()(zi, zj)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_hip_collaborative_gating(device, dtype, tol):
    from dvt_amd.models.collabgating import CollaborativeGating
    g = golden("fusion.npz")
    net = CollaborativeGating(compute_dtype=dtype)
    net.load_state_dict(_cg_weights(g))
    net = net.cuda()
    batch = [[[t.cuda() for t in experts] for experts in scenes] for scenes in _cg_inputs(g)]
    y = net(batch)
    assert y.shape == (2, 2, 1024) and rel_l2(y, T(g["cg_out"])) < tol
    y.backward(T(g["cg_gy"]).to(y.dtype).cuda())
    check_grad_digest(g, {k: p.grad for k, p in net.named_parameters()}, 20 * tol, "collabgating")


# ------------------------------------------------------------------ NT-Xent across ranks (gloo, CPU plumbing only)
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _gather_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvt_amd.models.losses.ntxent import _GatherRows
    x = (torch.arange(6, dtype=torch.float32).view(3, 2) + 10 * rank).requires_grad_(True)
    y = _GatherRows.apply(x, None)
    w = torch.arange(y.numel(), dtype=torch.float32).view_as(y)
    (y * w).sum().backward()
    out.put((rank, y.detach().numpy(), x.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ntxent_gather_rows_forward_and_local_gradient():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, y, gx = out.get(timeout=120)
        res[r] = (y, gx)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = np.concatenate([np.arange(6, dtype=np.float32).reshape(3, 2), np.arange(6, dtype=np.float32).reshape(3, 2) + 10])
    w = np.arange(12, dtype=np.float32).reshape(6, 2)
    for r in (0, 1):
        assert np.array_equal(res[r][0], full)
        assert np.array_equal(res[r][1], w[3 * r: 3 * r + 3])       # each rank keeps its own slice of the gradient
