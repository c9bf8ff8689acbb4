#!/usr/bin/env python3
"""Headline benchmark: clips/sec, forward + backward (+ gradient all-reduce under DP
and the fused AdamW step), ViViT metric shape of BASELINE.json
(B=8 per GPU, T=32, 3x224x224, bf16, d=512, 4+4 layers, 8 heads), synthetic data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel family (the MFMA GEMM, forward layout) measured live with
                HIP events on the launch stream inside the timed steps: algorithmic
                FLOPs of its launches / their summed duration, vs the dense bf16 MFMA peak.
  cpu_baseline  the CPU oracle (the reference's arithmetic restated, pinned to the
                reference by tests/golden) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def algorithmic_flops_per_clip(T, n_tok, d, heads, dh, depth, patch_dim, n_patch):
    """2 flop/MAC, forward; backward = 2x forward."""
    inner = heads * dh

    def layer(tokens_per_seq, seqs):
        m = tokens_per_seq * seqs
        qkv = 2 * m * d * 3 * inner
        qk = 2 * seqs * heads * tokens_per_seq ** 2 * dh
        pv = qk
        proj = 2 * m * inner * d
        ff = 4 * m * d * 4 * d
        return qkv + qk + pv + proj + ff

    fwd = 2 * T * n_patch * patch_dim * d + depth * layer(n_tok, T) + depth * layer(T + 1, 1)
    return fwd, 3 * fwd


def cls_fold_flag(B, T, n_tok, d, heads, dh, cdt):
    """Whether the space stack's last layer takes the folded single-query form: the predicate functional._AttnBlockCls uses
    (rows >= CLS_FOLD_MIN_ROWS, 16-bit element type, shapes the folded kernels and their head-wise helpers accept)."""
    from dvt_amd import functional as F
    probe = torch.empty((B * T, n_tok, d), dtype=cdt, device="cuda")
    return bool(F.cls_fold_taken(probe, heads, dh))


def executed_flops_per_clip(T, n_tok, d, heads, dh, depth, patch_dim, n_patch, pool_cls=True, fold_kv=True):
    """FLOPs the build actually launches: the reference reads only row 0 of the space transformer's output
    (vit.py:119-120) and, under pool == 'cls', of the temporal one (:126), so in the last layer of each stack the query /
    attention / output projection / feed-forward run on one row per sequence (keys and values on all rows).  fold_kv:
    the space stack's last layer runs with the K / V projections folded into its one query (csrc/attention_cls.hip:
    heads dot products and heads weighted sums of d elements per row instead of the [rows, d] x [d, 2 inner] product)."""
    inner = heads * dh

    def layer(tokens_per_seq, seqs, rows_out, folded=False):
        m, q = tokens_per_seq * seqs, rows_out * seqs
        kv = 2 * m * d * 2 * inner
        if folded:
            kv = 2 * m * d * 2 * heads + 3 * 2 * q * d * inner      # per row: scores + weighted sums; per sequence: r, o
        qp = 2 * q * d * inner
        qk = 2 * heads * q * tokens_per_seq * dh
        proj = 2 * q * inner * d
        ff = 4 * q * d * 4 * d
        return kv + qp + 2 * qk + proj + ff

    fwd = 2 * T * n_patch * patch_dim * d
    fwd += (depth - 1) * layer(n_tok, T, n_tok) + layer(n_tok, T, 1, folded=fold_kv)
    fwd += (depth - 1) * layer(T + 1, 1, T + 1) + layer(T + 1, 1, 1 if pool_cls else T + 1)
    return fwd, 3 * fwd


class EventProfiler:
    """HIP-event pairs around tagged launches on the current (launch) stream."""

    def __init__(self):
        self.open = {}
        self.records = {}

    def begin(self, key, flops):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.open[key] = (e0, flops)

    def end(self, key):
        e0, flops = self.open.pop(key)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.setdefault(key, []).append((e0, e1, flops))

    def calibrate(self, launch_noop, n=64):
        """A HIP event pair costs device time of its own (signal + barrier packets around the bracketed launch); the
        bracket of a no-op kernel measures that cost, which ``summary`` removes from every record.  The no-op's own
        run time (a single lane, a few microseconds of launch latency) is kept in, i.e. the correction is conservative."""
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch_noop(); e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) for a, b in ts)
        self.overhead_ms = max(0.0, v[len(v) // 2] - 0.004)      # minus ~4 us: the no-op kernel itself
        return self.overhead_ms

    overhead_ms = 0.0

    def summary(self):
        out = {}
        for key, recs in self.records.items():
            ms = sum(max(a.elapsed_time(b) - self.overhead_ms, 1e-3) for a, b, _ in recs)
            fl = sum(f for _, _, f in recs)
            out[key] = (ms, fl, len(recs))
        return out


def cpu_baseline(cfg, batch=8, timed_steps=3):
    """The oracle (the reference's arithmetic restated, pinned to the reference by tests/golden) timed on this host:
    forward + BCE + backward of the metric workload at B = 8, 1 warm-up + ``timed_steps`` timed
    steps, on a pinned 16 threads."""
    from oracle import clip_path as O
    from dvt_amd.models.vit import ViViT
    torch.manual_seed(1130)
    net = ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["T"], dim=cfg["d"], depth=cfg["depth"],
                heads=cfg["heads"], dim_head=cfg["dh"])
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    gen = torch.Generator().manual_seed(1130)
    x = torch.randn(batch, cfg["T"], 3, cfg["image"], cfg["image"], generator=gen)
    y = (torch.rand(batch, cfg["classes"], generator=gen) < 0.2).float()
    y[:, 0] = 1.0

    def step(n):
        t0 = time.perf_counter()
        O.vivit_step_fwd_bwd(x[:n], y[:n], P, patch=cfg["patch"], depth=cfg["depth"], heads=cfg["heads"])
        return time.perf_counter() - t0

    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    # A PINNED thread count: 16 threads (fewer when the process may use fewer).  The GPU hosts of the pool
    # are shared 256-thread machines: their full count is oversubscribed by an order of magnitude (measured earlier: 2.9 / 5.8 /
    # 106 s per clip at 64 / 128 / 256 threads) and the best count of a sweep moved between 8 and 32 from run to run, which
    # moved the baseline by 20 %; 16 is where the round-3 / round-4 sweeps landed and twice the survey container's 8 cores.
    best = min(16, cores)
    torch.set_num_threads(best)
    step(batch)                             # warm-up at the full batch
    ts = sorted(step(batch) for _ in range(timed_steps))
    dt = ts[len(ts) // 2]
    return {"value": round(batch / dt, 4), "unit": "clips/s", "cores": best, "kind": "port",
            "sample": f"oracle fp32 fwd+BCE+bwd, metric shape B={batch}, 1 warm-up + {timed_steps} timed steps, median "
                      f"{dt:.2f} s/step, {best} threads",
            "oracle": "pure-torch fp32 restatement of src/models/vit.py (oracle/clip_path.py), pinned by tests/golden",
            "host_logical_cpus": cores, "threads": f"pinned at {best} of a shared host's {cores}",
            "survey_cross_check": "the imported reference itself: 0.69 clips/s at B=8 on 8 cores, fp32 (BASELINE.md section 2)"}


WORKLOADS = {
    "vivit": "ViViT metric shape (SURVEY 8 'M'; BASELINE configs[2]/[3] transformer): "
             "B=8/GPU, T=32, 3x224x224, patch 16, d=512, depth 4+4, heads 8, dim_head 64; "
             "step = fwd + BCE + bwd + DP grad all-reduce + fused AdamW",
    "pyramid": "BASELINE configs[2]: per-frame ResNet-18 3-scale pyramid (train-mode BatchNorm) -> FPN lateral tokens "
               "(196/frame) -> the metric-shape transformer; B=8/GPU, T=32, 3x224x224; "
               "step = fwd + BCE + bwd + DP grad all-reduce + fused AdamW",
    "frametransformer": "reference default FrameTransformer(model='vid') (frame_transformer.py:83-121,192-210): R(2+1)D-18 "
                        "encoder on [B, 14, 12, 3, 112, 112] chunks (learnable pixel-space CLS chunk included), 896-d tokens, "
                        "4-layer post-norm encoder (dropout 0.5, training mode), BCE; B=2/GPU (config.yaml:2); unit = samples",
    "longclip": "BASELINE configs[4]: ViViT d=512, depth 4+4, T=64, 3x288x288 (325 tokens/frame), fp16 kernels + dynamic loss "
                "scaling (device-side), activation checkpointing (one saved activation per block); B=8/GPU",
    "crossmodal": "BASELINE configs[3]: configs[2] + 32 audio tokens (128-d), cross-attention block (video queries, "
                  "audio keys/values), distillation token + head, loss = BCE + hard-label CE; B=8/GPU, T=32, 3x224x224",
}


def newest_pmc_traffic():
    """Per-launch HBM bytes of the GEMM families from the newest committed PMC profile (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes cannot run inside this process; tools/run_profiles.sh collects them with this same command)."""
    import glob
    import re
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))
                    if re.fullmatch(r"r\d+_pmc_traffic\.json", os.path.basename(f))),
                   key=lambda f: int(re.search(r"r(\d+)_pmc_traffic", f).group(1)))
    if not files:
        return {}, None
    try:
        with open(files[-1]) as fh:
            return json.load(fh)["families"], os.path.relpath(files[-1], ROOT)
    except Exception:
        return {}, None


def full_k_reference(ops):
    """The dominant kernel on a product without the metric shapes' structure (4096^3 bf16, 64 k-tiles per tile, one round of
    256 tiles, negligible epilogue), measured live on this device on random data: what the same main loop sustains when the
    K = 512 effects DESIGN 4.1 lists (epilogue 33 %, prologue 7 %, partial rounds) are absent.  On random bf16 data the chip
    holds 1.9-2.0 GHz in such a loop (MI355X_MICROARCH.md, DVFS give-back (1)), so 2.5 PFLOP/s is not reachable by any kernel."""
    try:
        n = 4096
        a = torch.randn(n, n, device="cuda").to(torch.bfloat16)
        b = torch.randn(n, n, device="cuda").to(torch.bfloat16)
        c = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
        run = lambda: ops.gemm(a, b, n, n, n, a_kmajor=True, b_kmajor=True, lda=n, ldb=n, out=c)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tf = 2.0 * n ** 3 / (us * 1e-6) / 1e12
        return {"shape": "4096^3 bf16, A and B k-major", "us_per_launch": round(us, 1), "achieved": round(tf, 1),
                "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4)}
    except Exception as e:      # informational: never takes the line down
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def build_workload(args, workload, rank, comm):
    """Model, flat parameter store, synthetic batch and the step closure of one workload."""
    from dvt_amd import functional as F
    from dvt_amd.dp import FlatParameters
    from dvt_amd.models.vit import ViViT

    cfg = dict(image=224, patch=16, classes=19, T=32, d=512, depth=4, heads=8, dh=64)
    cdt = torch.bfloat16
    if workload == "longclip":
        cfg.update(image=288, T=64)
        cdt = torch.float16
    elif args.dtype == "fp16":
        cdt = torch.float16
    B = args.batch
    torch.manual_seed(1130)                                       # src/main.py:25
    if workload == "frametransformer":
        from dvt_amd.models.frame_transformer import FrameTransformer
        if B == 8:
            B = 2                                              # config.yaml:2
        net = FrameTransformer(batch_size=B, seq_len=13, cls=1, model="vid", opt="adamW", learning_rate=5e-6,
                               weight_decay=0.09, momentum=0.005).cuda().train()
    elif workload in ("vivit", "longclip"):
        net = ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["T"], dim=cfg["d"], depth=cfg["depth"],
                    heads=cfg["heads"], dim_head=cfg["dh"], compute_dtype=cdt,
                    activation_checkpointing=workload == "longclip").cuda().train()
    else:
        from dvt_amd.models.pyramid_vivit import PyramidViViT
        cm = workload == "crossmodal"
        net = PyramidViViT(cfg["image"], cfg["classes"], cfg["T"], dim=cfg["d"], depth=cfg["depth"], heads=cfg["heads"],
                           dim_head=cfg["dh"], audio_tokens=32 if cm else 0, audio_dim=128, distill=cm,
                           compute_dtype=torch.bfloat16).cuda().train()
    rdt = {"fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[args.grad_dtype] if comm is not None else None
    flat = FlatParameters(net, bucket_mb=args.bucket_mb, compute_dtype=cdt, comm=comm, grad_reduce_dtype=rdt)
    flat.broadcast_parameters(0)
    flat.sync_compute_copy()
    gen = torch.Generator().manual_seed(1130 + rank)
    if workload == "frametransformer":
        x = torch.randn(B, 13, 12, 3, 112, 112, generator=gen).cuda()     # MMX_Light_dl.py:286 batch contract
    else:
        x = torch.randn(B, cfg["T"], 3, cfg["image"], cfg["image"], generator=gen).to(cdt).cuda()
    y = (torch.rand(B, cfg["classes"], generator=gen) < 0.2).float()
    y[:, 0] = 1.0
    y = y.cuda()
    gloss = torch.full((), flat.loss_scale, device="cuda")
    if cdt == torch.float16:
        gloss = flat.enable_loss_scaling(init_scale=4096.0, growth_interval=1000)     # device-resident seed
    audio = None
    if workload == "crossmodal":       # one 128-d VGGish-style vector per 1-s chunk (SURVEY 8d synthetic inputs)
        audio = torch.randn(B, 32, 128, generator=gen).to(torch.bfloat16).cuda()

    def fwd_bwd():
        flat.zero_grad()
        if workload in ("vivit", "longclip"):
            loss, _ = net.loss(x, y)                    # BCEWithLogits(net(x), y), head + loss in one launch
        elif workload == "frametransformer":
            loss = net.training_step((y, None, x), 0)
        else:
            loss = net.training_step((y, x, audio) if audio is not None else (y, x))
        loss.backward(gloss)
        return loss

    def update():
        flat.adamw_step(lr=5e-6, weight_decay=0.09)              # config.yaml:3,12

    def step():
        loss = fwd_bwd()
        flat.finish_backward()
        update()
        return loss

    return dict(cfg=cfg, cdt=cdt, B=B, net=net, flat=flat, step=step, fwd_bwd=fwd_bwd, update=update)


def collective_capture_ok(comm):
    """Can this RCCL build's all-reduce be recorded in a hipGraph and replayed with the right result?  (Checked on a small
    buffer on every rank before the step is captured; a refusal or a wrong sum selects the segmented form.)"""
    try:
        world, rank = comm.world, comm.rank
        t = torch.full((4096,), float(rank + 1), device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            comm.all_reduce_async(t).wait()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        want = world * (world + 1) / 2.0
        if float(t[0]) != want:
            return False, f"eager all-reduce returned {float(t[0])}, expected {want}"
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            comm.all_reduce_async(t).wait()
        for _ in range(2):
            t.fill_(float(rank + 1))
            g.replay()
            torch.cuda.synchronize()
            if float(t[0]) != want or float(t[-1]) != want:
                return False, f"replayed all-reduce returned {float(t[0])}, expected {want}"
        return True, ""
    except Exception as e:      # capture refused
        return False, f"{type(e).__name__}: {e}"[:200]


def profile_pass(ops, step, ms_per_step, steps):
    """A separate short pass with a HIP-event pair around every tagged launch (so that the records do not perturb the
    headline timing).  The host needs longer to issue a step than the GPU to run it when every launch is bracketed; a
    device-side delay in front of each profiled step lets the host run ahead, so the brackets see device time only.
    -> ({key: (ms, units, launches)}, profiled steps, bracket overhead in us)."""
    prof = EventProfiler()
    ops.set_profiler(prof)
    nprof = max(1, min(3, steps))
    host_ms = max(20.0, 4.0 * ms_per_step)
    ops.device_delay(20000)
    bracket_us = prof.calibrate(lambda: ops.device_delay(0)) * 1e3
    for _ in range(nprof):
        for _ in range(int(host_ms // 200) + 1):
            ops.device_delay(int(min(host_ms, 200.0) * 1000))
        step()
    torch.cuda.synchronize()
    ops.set_profiler(None)
    return prof.summary(), nprof, bracket_us


def cnn_roofline(summ, nprof, workload):
    """Roofline of the per-frame CNN encoder's kernels inside a secondary workload (custom_resnet.py:19-22,100-109): the
    convolution families against the dense bf16 MFMA peak (algorithmic FLOPs 2 * pixels * Cout * kh * kw * Cin), the
    BatchNorm passes against the HBM peak (algorithmic bytes: every operand / result once per pass), and -- from the
    committed PMC profile of the same workload -- HBM bytes per launch against the algorithmic bytes."""
    fams = {}
    for key, (ms, fl, cnt) in summ.items():
        if key[0] != "conv":
            continue
        _, kind, M, N, K, nbytes = key
        if kind == "wgrad":                # key = (.., K_out = kh*kw*Cin, Cout, rows): class by output channels
            name = f"implicit weight gradient, Cout {'<= 64' if N <= 64 else '65..128' if N <= 128 else '> 128'} (split-K; the reduce rides in the layer's data-gradient launch)"
        elif kind == "halo3x3_c64":
            name = "conv3x3_c64 (LDS halo patch; forward and data gradient of layer 1)"
        elif kind == "halo3x3_stream":
            name = "conv3x3_stream (LDS halo patch, streamed weights; 64 -> 144 forward and 144 -> 64 data gradient of R(2+1)D layer 1)"
        elif kind == "halo3x3_c64_wgrad":
            name = "conv3x3_c64 weight gradient (LDS halo patches; one partial per workgroup, summed by the split-K reduce)"
        elif kind == "window3x1_wgrad":
            name = "conv3x1 temporal weight gradient 144 -> 64 (LDS sliding windows over all frames of a pixel segment)"
        elif kind == "stream3x1_bn_bwd":
            name = "conv3x1 temporal data gradient 64 -> 144 with the mid-plane BatchNorm backward in its epilogue (computed twice: sums, then the corrected gradient; the 144-plane gradient is never stored)"
        elif kind == "window3x1_fwd":
            name = "conv3x1 temporal forward 144 -> 64 (LDS sliding windows; the spatial half's BatchNorm + ReLU applied in the window)"
        elif kind == "stem7":
            name = "conv_stem7 (7x7 / 2 stem on 3-channel frames: LDS halo patch of the pixel-pair map, weights in registers)"
        elif kind == "window3x1_c64":
            name = "conv3x1 temporal forward and data gradient 64 -> 64 of the stem (LDS sliding windows, 16-pixel segments)"
        else:
            name = f"implicit forward / data gradient, Cout {'<= 64' if N <= 64 else '65..128' if N <= 128 else '> 128'}"
        f = fams.setdefault(name, [0.0, 0.0, 0, 0.0, kind])
        f[0] += ms; f[1] += fl; f[2] += cnt; f[3] += float(nbytes) * cnt
    pmc = {}
    pfile = None
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc_traffic.json")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    if files:
        try:
            with open(files[-1]) as fh:
                pmc, pfile = json.load(fh)["families"], os.path.relpath(files[-1], ROOT)
        except Exception:
            pmc = {}
    pmc_key = {"implicit": "conv_implicit", "wgrad": "conv_wgrad", "halo3x3_c64": "conv3x3_c64",
               "halo3x3_c64_wgrad": "conv3x3_c64_wgrad", "halo3x3_stream": "conv3x3_stream",
               "window3x1_wgrad": "conv3x1_wgrad", "window3x1_fwd": "conv3x1_fwd", "window3x1_c64": "conv3x1_c64",
               "stream3x1_bn_bwd": "conv3x1_stream_bn_bwd", "stem7": "conv_stem7"}
    out = {"bound": "mfma", "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "conv_families": {}, "hbm_kernels": {},
           "traffic_source": pfile}
    for name, (ms, fl, cnt, nb, kind) in sorted(fams.items(), key=lambda kv: -kv[1][0]):
        tf = fl / (ms * 1e-3) / 1e12
        traffic = (pmc.get(pmc_key.get(kind, kind), {}) or {}).get("hbm_bytes_corrected")
        if traffic and kind == "stream3x1_bn_bwd":
            traffic *= 2                     # one operator = two kernel launches (sums pass + apply pass); the family averages both
        alg_kind = sum(v[3] for v in fams.values() if v[4] == kind) / max(1, sum(v[2] for v in fams.values() if v[4] == kind))
        # the same launches against the HBM peak on their ALGORITHMIC bytes: a family of narrow outputs (Cout <= 64 at a few
        # hundred k-columns: ~64 flops per byte) is bounded by its bytes, not by the MFMA pipe -- `bound` names the roof the
        # family is nearer to and `roof_frac` its fraction of that one
        gbs = nb / (ms * 1e-3) / 1e9
        mf, hf = tf / MFMA_PEAK_TFLOPS, gbs / HBM_PEAK_GBS
        out["conv_families"][name] = {"ms_per_step": round(ms / nprof, 3), "launches_per_step": cnt // nprof,
                                      "achieved": round(tf, 1), "frac": round(mf, 4),
                                      "algorithmic_MB_per_launch": round(nb / cnt / 1e6, 1),
                                      "algorithmic_GBps": round(gbs, 1), "hbm_frac": round(hf, 4),
                                      "bound": "hbm" if hf > mf else "mfma", "roof_frac": round(max(mf, hf), 4),
                                      "traffic_ratio": round(traffic / alg_kind, 3) if traffic else None}
    agg = {}
    for key, (ms, units, cnt) in summ.items():
        if key[0] == "hbm" and key[1].startswith("bn_"):
            a = agg.setdefault(key[1], [0.0, 0.0, 0])
            a[0] += ms; a[1] += units; a[2] += cnt
    for name, (ms, units, cnt) in agg.items():
        out["hbm_kernels"][name] = {"ms_per_step": round(ms / nprof, 3), "launches_per_step": cnt // nprof,
                                    "algorithmic_GBps": round(units / (ms * 1e-3) / 1e9, 1),
                                    "frac_of_hbm_peak": round(units / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
    if fams:
        dom = max(fams.values(), key=lambda v: v[0])
        out["achieved"] = round(dom[1] / (dom[0] * 1e-3) / 1e12, 1)
        out["frac"] = round(out["achieved"] / MFMA_PEAK_TFLOPS, 4)
        out["kernel"] = [k for k, v in fams.items() if v is dom][0]
    return out


def rebuild_communicator(comm):
    """A capture that failed inside a collective leaves the communicator and its side stream in an undefined state."""
    from dvt_amd.dp import Communicator
    try:
        comm.destroy()
    except Exception:
        pass
    torch.cuda.synchronize()
    _LIVE["comm"] = Communicator.from_torch_distributed()
    return _LIVE["comm"]


_LIVE = {"comm": None}      # the communicator main() destroys at exit (a rebuild replaces it)

# Distributed watchdog: a collective that some rank never joins hangs every rank silently.  The limit counts time since the
# last PROGRESS point (progress() re-arms it: after the rendezvous, each warm-up / timed block, each workload), not since
# the start, so a long legitimate run (large --steps, the secondaries) is not killed; it is cancelled before the final print.
_WATCHDOG = {"limit": 0.0, "rank": 0, "timer": None, "where": ""}


def progress(where=""):
    import threading
    if _WATCHDOG["limit"] <= 0:
        return
    if _WATCHDOG["timer"] is not None:
        _WATCHDOG["timer"].cancel()
    _WATCHDOG["where"] = where
    if where is None:                  # cancel only
        _WATCHDOG["timer"] = None
        return

    def _give_up():
        sys.stderr.write(f"[bench] rank {_WATCHDOG['rank']}: no progress for {_WATCHDOG['limit']:.0f} s after "
                         f"'{_WATCHDOG['where']}' (a collective is stuck?) -- aborting\n")
        sys.stderr.flush()
        os._exit(124)

    t = threading.Timer(_WATCHDOG["limit"], _give_up)
    t.daemon = True
    t.start()
    _WATCHDOG["timer"] = t


def timed_steps(run, steps, barrier):
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]   # per-step spread (p10 / median / p90)
    t0 = time.perf_counter()
    evs[0].record()
    loss = None
    for i in range(steps):
        loss = run()
        evs[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    return elapsed, per_step, loss


def run_workload(args, workload, rank, world, use_dist, comm, *, steps, warmup, roofline, secondary=False):
    """One workload, timed per the driver's contract.  Returns the JSON dict (rank 0) or None."""
    from dvt_amd import ops
    from dvt_amd.graph import capture_step, capture_step_segments
    W = build_workload(args, workload, rank, comm)
    cfg, cdt, B, flat, step = W["cfg"], W["cdt"], W["B"], W["flat"], W["step"]

    def barrier():
        if use_dist:
            dist.barrier()

    # Launch form.  One GPU: the whole step (fwd + loss + bwd + AdamW + weight cast) is one captured hipGraph.  Data
    # parallel: the SAME single graph, with the bucketed RCCL all-reduces (dvt_comm_allreduce on a side stream, forked
    # and joined through events) recorded inside it -- the host launches one graph per step whatever N is.  If this RCCL
    # build cannot be captured: two graphs around one eagerly launched all-reduce; last resort: eager launches.
    launch, note = "eager", ""
    run = step
    if not args.no_graph:
        if not use_dist:
            replay, static_loss = capture_step(step, warmup=2)
            launch = "hipGraph replay"
        else:
            ok, why = (True, "") if comm is None else collective_capture_ok(comm)
            if comm is not None:
                # every rank must take the same launch form (they enqueue different collective sequences otherwise):
                # MIN over the group; a rank whose capture failed part-way re-creates its communicator before reuse
                flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag) == 0.0:             # creating a communicator is itself collective: every rank rebuilds
                    comm = W["flat"].comm = rebuild_communicator(comm)
                    if ok:
                        why = "another rank could not capture the collective"
                ok = float(flag) == 1.0
            if comm is not None and ok:
                replay, static_loss = capture_step(step, warmup=2)
                launch = "hipGraph replay (bucketed RCCL all-reduce captured inside the step graph)"
            elif comm is not None:
                flat.defer_exchange = True

                def fwd_bwd_local():
                    loss = W["fwd_bwd"]()
                    flat.finish_backward(exchange=False)
                    return loss

                replay, static_loss = capture_step_segments(fwd_bwd_local, flat.exchange_all, W["update"], warmup=2)
                launch, note = "two hipGraphs around one eager RCCL all-reduce", why
            else:
                replay = None            # torch.distributed collectives of a non-RCCL backend: eager
        if launch != "eager":
            def run():
                replay()
                return static_loss

    progress(f"{workload}: capture")
    for _ in range(warmup):
        run()
    progress(f"{workload}: warm-up")
    elapsed, per_step, loss = timed_steps(run, steps, barrier)
    progress(f"{workload}: timed steps")
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.detach())
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]

    extra = {}
    if use_dist and not secondary:
        # the same step launched kernel by kernel from the host (what round 1 did under data parallelism)
        k = max(3, min(10, steps))
        for _ in range(2):
            step()
        e_elapsed, _, _ = timed_steps(step, k, barrier)
        extra["eager_ms_per_step"] = round(e_elapsed / k * 1e3, 3)
        progress(f"{workload}: eager steps")
        if comm is not None:
            # One traced eager step: when does each bucket's all-reduce start and end relative to the END of backward
            # (negative start = launched while backward was still running, i.e. overlapped)?  Side-stream timing events.
            comm.trace, comm.marks = [], {}
            step()
            torch.cuda.synchronize()
            end_bwd = comm.marks.get("backward_end")
            if end_bwd is not None:
                extra["bucket_timeline"] = [{"wire_bytes": int(nb), "start_ms_after_backward_end": round(end_bwd.elapsed_time(e0), 3),
                                             "end_ms_after_backward_end": round(end_bwd.elapsed_time(e1), 3),
                                             "ms": round(e0.elapsed_time(e1), 3)} for nb, e0, e1 in comm.trace]
            comm.trace, comm.marks = None, {}
        extra["graph_ms_per_step"] = round(elapsed / steps * 1e3, 3)
        if comm is not None and launch.startswith("hipGraph replay"):
            # exposed (un-overlapped) exchange time: the captured step with and without its collectives
            flat.exchange_enabled = False
            replay0, _ = capture_step(step, warmup=1)
            for _ in range(2):
                replay0()
            n_elapsed, _, _ = timed_steps(lambda: replay0(), k, barrier)
            flat.exchange_enabled = True
            extra["graph_ms_per_step_without_exchange"] = round(n_elapsed / k * 1e3, 3)
            extra["exposed_allreduce_ms_per_step"] = round(max(0.0, elapsed / steps - n_elapsed / k) * 1e3, 3)
            extra["allreduce_bytes_per_step"] = int(flat.total * (2 if flat.grad16 is not None else 4))

    # optimizer share of the step (SURVEY 8d asks for it separately): fused AdamW + bf16 weight mirror on the flat buffers
    o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    o0.record()
    for _ in range(5):
        flat.adamw_step(lr=0.0, weight_decay=0.0)            # lr = 0: timing only, the weights stay put
    o1.record()
    torch.cuda.synchronize()
    optimizer_ms = o0.elapsed_time(o1) / 5

    # ---- roofline of the dominant kernel family, live HIP events (separate short pass so the
    #      event records do not perturb the headline timing)
    roof = None
    if roofline and workload in ("pyramid", "crossmodal", "frametransformer"):      # the CNN encoder's kernels
        summ, nprof, _ = profile_pass(ops, step, elapsed / steps * 1e3, steps)
        roof = cnn_roofline(summ, nprof, "frametransformer" if workload == "frametransformer" else "pyramid")
    elif roofline and workload == "longclip":       # configs[4]: the HBM-bandwidth report
        summ, nprof, _ = profile_pass(ops, step, elapsed / steps * 1e3, steps)
        roof = longclip_roofline(summ, nprof, round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    elif roofline:
        summ, nprof, bracket_us = profile_pass(ops, step, elapsed / steps * 1e3, steps)
        fam, hbm, big = {}, {}, {}
        for key in summ:                       # per kernel: its largest launch shape (the space-transformer one)
            if key[0] == "hbm" and (key[1] not in big or key[2] > big[key[1]][2]):
                big[key[1]] = key
        for name, key in big.items():
            ms, units, cnt = summ[key]
            hbm[name] = {"us_per_launch": round(ms * 1e3 / cnt, 1), "launches_per_step": cnt // nprof,
                         "algorithmic_MB_per_launch": round(units / cnt / 1e6, 1),
                         "algorithmic_GBps": round(units / (ms * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak": round(units / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
            if len(key) > 3:        # attention: the QK^T / PV (and backward) products are MFMA work -- report that roofline too
                tf = key[3] * cnt / (ms * 1e-3) / 1e12
                hbm[name]["mfma_TFLOPs"] = round(tf, 1)
                hbm[name]["frac_of_mfma_peak"] = round(tf / MFMA_PEAK_TFLOPS, 3)
                hbm[name]["mfma_ceiling_note"] = ("fused attention moves 4 N dh 2 B per (frame, head) for 4 N^2 dh flop: N/2 flop/B "
                                                  "= 98.5 at N=197 against a machine balance of 312, i.e. at most 31 % of the MFMA "
                                                  "peak at the HBM roofline (north_star asks 40 %)")
        for key, (ms, fl, cnt) in summ.items():
            if key[0] != "gemm":
                continue
            tag, ak, bk, M, N, K, epi, nbytes = key
            if M * N * K < (1 << 30):      # launch-bound temporal/head GEMMs: not this kernel's regime
                continue
            f = fam.setdefault((ak, bk), [0.0, 0.0, 0, 0.0])
            f[0] += ms; f[1] += fl; f[2] += cnt; f[3] += float(nbytes) * cnt
        names = {(1, 1): "gemm_dma_kernel<A k-major, B k-major> (forward Linear)",
                 (1, 0): "gemm_dma_kernel<A k-major, B mn-major> (data gradient; carries the weight gradients' split-K reduces in its grid tail)",
                 (0, 0): "gemm_dma_kernel<A mn-major, B mn-major> (weight gradient, split-K; its reduce rides in the data-gradient launch behind it)"}
        pmc, pmc_file = newest_pmc_traffic()
        pmc_key = {(1, 1): "fwd", (1, 0): "dgrad", (0, 0): "wgrad"}
        if fam:
            # dominant = the layout that carries the most algorithmic FLOPs (the three sit within a few percent of each
            # other in time; the data-gradient launches also carry the weight gradients' split-K reduces in their grid
            # tails -- time without FLOPs --, which would otherwise decide the choice from run to run)
            dom = max(fam, key=lambda k: (fam[k][1], fam[k][0]))
            ms, fl, cnt, nb = fam[dom]
            ach = fl / (ms * 1e-3) / 1e12
            traffic = (pmc.get(pmc_key.get(dom, ""), {}) or {}).get("hbm_bytes_corrected")
            alg = nb / cnt
            roof = {"bound": "mfma", "kernel": names.get(dom, str(dom)), "achieved": round(ach, 1),
                    "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic,
                    "traffic_source": f"committed profile {pmc_file} (rocprofv3 --pmc, 2*FETCH_SIZE + WRITE_SIZE per launch; "
                                      "FETCH_SIZE includes Infinity-Cache hits: an upper bound of the HBM reads)" if pmc_file else None,
                    "algorithmic_bytes_per_launch": int(alg),
                    "traffic_ratio": round(traffic / alg, 3) if traffic else None,
                    "launches": cnt // nprof, "avg_launch_us": round(ms * 1e3 / cnt, 1),
                    "dominant_by": "algorithmic FLOPs per step among the kernel's three operand layouts",
                    "families": {names.get(k, str(k)): {"ms_per_step": round(v[0] / nprof, 3),
                                                        "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                                                        "algorithmic_MB_per_launch": round(v[3] / v[2] / 1e6, 1)}
                                 for k, v in fam.items()},
                    "hbm_kernels": hbm,
                    "same_kernel_full_k": full_k_reference(ops),
                    "event_bracket_overhead_us": round(bracket_us, 1),
                    "timing_note": "HIP-event pair per launch on the launch stream, behind a device-side delay so that the host "
                                   "runs ahead; the measured cost of an empty bracket is subtracted from every record"}

    rank_devices = None
    if use_dist:                       # which device every rank ran on ("RCCL saw N ranks" is checkable from the line)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, f"rank {rank}: cuda:{torch.cuda.current_device()} "
                                             f"{torch.cuda.get_device_name()} pid {os.getpid()}")
    out = None
    if rank == 0:
        ms_step = elapsed / steps * 1e3
        clips = B * world * steps / elapsed
        n_tok = (cfg["image"] // cfg["patch"]) ** 2 + 1
        fwd, tot = algorithmic_flops_per_clip(cfg["T"], n_tok, cfg["d"], cfg["heads"], cfg["dh"], cfg["depth"],
                                              3 * cfg["patch"] ** 2, n_tok - 1)
        is_ft = workload == "frametransformer"
        exec_flops = 0.0
        if workload in ("vivit", "longclip"):
            fold = cls_fold_flag(B, cfg["T"], n_tok, cfg["d"], cfg["heads"], cfg["dh"], cdt)
            exec_flops = executed_flops_per_clip(cfg["T"], n_tok, cfg["d"], cfg["heads"], cfg["dh"], cfg["depth"],
                                                 3 * cfg["patch"] ** 2, n_tok - 1, fold_kv=fold)[1]
        out = {
            "metric": f"clips/sec fwd+bwd, B=8 T=32 3x224x224 {args.dtype}" if workload == "vivit" else
                      ("samples/sec fwd+bwd [frametransformer workload], B=2 x 14 chunks x 12 x 3x112x112" if is_ft else
                       f"clips/sec fwd+bwd [{workload} workload], B={B} T={cfg['T']} 3x{cfg['image']}x{cfg['image']}"),
            "value": round(clips, 2), "unit": "samples/s" if is_ft else "clips/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "fp16" if cdt == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": WORKLOADS[workload],
                       "global_batch": B * world, "parallelism": f"dp{world}", "params_M": round(flat.total / 1e6, 2)},
            "launch": launch,
            "model_tflops": round(tot * B * world / (elapsed / steps) / 1e12, 1)
            if workload in ("vivit", "longclip") else None,
            "model_mfma_frac": round(tot * B / (elapsed / steps) / 1e12 / MFMA_PEAK_TFLOPS, 4)
            if workload in ("vivit", "longclip") else None,
            # model_tflops prices the step at the REFERENCE's algorithmic FLOPs (dense last layers); executed_tflops at
            # what is launched (last layer of each stack on the CLS rows only)
            "executed_tflops": round(exec_flops * B * world / (elapsed / steps) / 1e12, 1)
            if workload in ("vivit", "longclip") else None,
            # the hardware fraction: launched FLOPs per second per GPU against the dense MFMA peak
            "executed_mfma_frac": round(exec_flops * B / (elapsed / steps) / 1e12 / MFMA_PEAK_TFLOPS, 4)
            if workload in ("vivit", "longclip") else None,
            "final_loss": round(final_loss, 5),
            "step_ms": {"p10": round(pct(0.1), 3), "median": round(pct(0.5), 3), "p90": round(pct(0.9), 3)},
            "optimizer_ms_per_step": round(optimizer_ms, 3),
            "peak_hbm_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
        }
        if note:
            out["launch_note"] = note
        if use_dist:
            out["gradient_exchange"] = {"world": comm.world if comm is not None else dist.get_world_size(),
                                        "rank_devices": rank_devices,
                                        "dtype": args.grad_dtype if comm is not None else "fp32", "bucket_mb": args.bucket_mb,
                                        "through": "dvt_comm_allreduce (RCCL behind the C ABI)" if comm is not None
                                        else f"torch.distributed ({args.backend})", **extra}
        if cdt == torch.float16:
            out["loss_scale"] = float(flat.scale_dev)
        if roof is not None:
            out["roofline"] = roof
    return out


def release_gpu_memory():
    """Between workloads: the previous one's graphs, activations and flat buffers are garbage once run_workload returned."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


def launch_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment), relay rank 0's JSON line, exit non-zero when any rank fails.  This
    parent never touches the GPU (no HIP call before or after the spawn; a process that has initialised the GPU must not
    be re-exec'd), it only waits."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sk:                    # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DVT_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    box = {"line": None}

    def relay():                                   # rank 0 prints the ONE JSON line; anything else it prints is passed on
        for ln in procs[0].stdout:
            if ln.lstrip().startswith("{"):
                box["line"] = ln.strip()
            else:
                sys.stderr.write(ln)

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True                          # a dead rank leaves the others in a collective: end exactly those PIDs
            for q in procs:
                if q.poll() is None:
                    q.send_signal(signal.SIGTERM)
            time.sleep(2.0)
            for q in procs:
                if q.poll() is None:
                    q.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    th.join(timeout=5.0)
    line = box["line"]
    if failed or any(rcs) or line is None:
        raise SystemExit(f"[bench] rank exit codes {rcs}; JSON line {'missing' if line is None else 'present'}")
    print(line, flush=True)

DETAIL_DEFAULT = os.path.join("profiles", "r06_bench_detail.json")
LINE_LIMIT = 6000      # the driver keeps the last 8 KB of stdout: the ONE line must fit with room to spare


def _short(s, n=110):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def longclip_roofline(summ, nprof, peak_gib):
    """BASELINE configs[4] / SURVEY 8(d) config 5: HBM GB/s of the streaming kernels of the long-clip step (LayerNorm,
    pos-add / CLS assembly, patchify, the GELU-epilogue GEMM's own traffic) and of the N = 325 attention pair, each as
    algorithmic bytes per launch / live HIP-event launch time, against the HBM peak; peak activation memory."""
    from dvt_amd import _lib as L
    ker = {}
    for key, (ms, units, cnt) in summ.items():
        if key[0] == "hbm":
            k = ker.setdefault(key[1], {"ms": 0.0, "bytes": 0.0, "n": 0, "flops": 0.0, "big": 0})
            if key[2] >= k["big"]:                  # per kernel: report its largest launch shape (the space stack's)
                if key[2] > k["big"]:
                    k.update(ms=0.0, bytes=0.0, n=0, flops=0.0, big=key[2])
                k["ms"] += ms; k["bytes"] += units; k["n"] += cnt
                if len(key) > 3:
                    k["flops"] += key[3] * cnt
        elif key[0] == "gemm" and key[6] == L.EPI_GELU and key[3] * key[4] * key[5] >= (1 << 30):
            k = ker.setdefault("ff1_gemm_gelu_epilogue", {"ms": 0.0, "bytes": 0.0, "n": 0, "flops": 0.0, "big": 0})
            k["ms"] += ms; k["bytes"] += float(key[7]) * cnt; k["n"] += cnt; k["flops"] += units
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": {}, "peak_activation_GiB": peak_gib}
    for name, k in sorted(ker.items(), key=lambda kv: -kv[1]["ms"]):
        if k["ms"] <= 0:
            continue
        gbs = k["bytes"] / (k["ms"] * 1e-3) / 1e9
        e = {"us": round(k["ms"] * 1e3 / k["n"], 1), "n": k["n"] // nprof, "GBps": round(gbs), "frac": round(gbs / HBM_PEAK_GBS, 3)}
        if k["flops"]:
            e["mfma_frac"] = round(k["flops"] / (k["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 3)
        out["kernels"][name] = e
    stream = {n: k for n, k in ker.items() if not k["flops"] and k["ms"] > 0}
    if stream:                                       # the dominant pure-streaming kernel carries achieved / frac
        dom = max(stream, key=lambda n: stream[n]["ms"])
        out["kernel"] = dom
        out["achieved"] = out["kernels"][dom]["GBps"]
        out["frac"] = out["kernels"][dom]["frac"]
        out["traffic"] = None
    return out


def compact_line(out):
    """The ONE line the driver records: every workload's value / ms_per_step, the dominant kernel's roofline and the CPU
    baseline, short enough for the driver's 8 KB tail; everything else goes to the detail file."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in out}
    cfg = dict(out.get("config", {}))
    cfg["workload"] = _short(cfg.get("workload", ""), 118)
    line["config"] = cfg
    r = out.get("roofline")
    if r:
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                "traffic_ratio", "launches", "avg_launch_us", "peak_activation_GiB")
        line["roofline"] = {k: (_short(r[k], 72) if isinstance(r[k], str) else r[k]) for k in keep if k in r}
        fk = (r.get("same_kernel_full_k") or {}).get("frac")
        if fk is not None:
            line["roofline"]["same_kernel_4096_cubed_frac"] = fk
        if "kernels" in r:
            line["roofline"]["kernels"] = r["kernels"]
    for k in ("launch", "executed_mfma_frac", "model_mfma_frac", "final_loss", "optimizer_ms_per_step", "peak_hbm_GiB", "step_ms"):
        if out.get(k) is not None:
            line[k] = _short(out[k], 60) if isinstance(out[k], str) else out[k]
    ge = out.get("gradient_exchange")
    if ge:
        line["gradient_exchange"] = {k: ge[k] for k in ("world", "dtype", "bucket_mb", "through", "eager_ms_per_step",
                                                        "graph_ms_per_step", "graph_ms_per_step_without_exchange",
                                                        "exposed_allreduce_ms_per_step", "allreduce_bytes_per_step") if k in ge}
        line["gradient_exchange"]["devices"] = sorted({d.split(": ", 1)[1].rsplit(" pid", 1)[0] for d in ge.get("rank_devices") or []})
    sec = out.get("secondary")
    if sec:
        line["secondary"] = {}
        for wl, s in sec.items():
            if "error" in s:
                line["secondary"][wl] = {"error": _short(s["error"], 100)}
                continue
            e = {k: s[k] for k in ("value", "unit", "ms_per_step", "dtype", "peak_hbm_GiB") if k in s}
            e["launch"] = "graph" if str(s.get("launch", "")).startswith("hipGraph") else "eager"
            rr = s.get("roofline")
            if rr and "kernels" in rr:               # longclip: the HBM report configs[4] asks for
                e["roofline"] = {k: rr[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernels",
                                                    "peak_activation_GiB") if k in rr}
            elif rr:
                e["roofline"] = {"bound": rr.get("bound"), "kernel": _short(rr.get("kernel", ""), 60), "frac": rr.get("frac")}
                fams = rr.get("conv_families") or {}
                if fams:
                    e["roofline"]["conv_frac_min"] = min(f["frac"] for f in fams.values())
                    e["roofline"]["conv_roof_frac_min"] = min(f.get("roof_frac", f["frac"]) for f in fams.values())
                    tr = [f["traffic_ratio"] for f in fams.values() if f.get("traffic_ratio")]
                    e["roofline"]["traffic_ratio_max"] = max(tr) if tr else None
                bn = rr.get("hbm_kernels") or {}
                if bn:
                    e["roofline"]["batchnorm_ms"] = round(sum(b["ms_per_step"] for b in bn.values()), 3)
            line["secondary"][wl] = e
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: (_short(cb[k], 118) if isinstance(cb[k], str) else cb[k])
                                for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
    return line


def emit(out, detail_path):
    """Write the full record to the detail file (tracked under profiles/ when the run is the builder's own) and print the
    compact line, which names that file."""
    line = compact_line(out)
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.join(ROOT, detail_path)) or ".", exist_ok=True)
            with open(os.path.join(ROOT, detail_path), "w") as fh:
                json.dump(out, fh, indent=1)
            line["detail"] = detail_path
        except OSError as e:                          # a read-only tree must not take the line down
            line["detail"] = f"not written ({type(e).__name__})"
    txt = json.dumps(line, separators=(",", ":"))
    for drop in ("step_ms", "optimizer_ms_per_step", "final_loss", "launch"):      # never expected; the limit is a contract
        if len(txt) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        txt = json.dumps(line, separators=(",", ":"))
    print(txt, flush=True)
    return txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bounded secondary workloads (pyramid, crossmodal, "
                    "longclip, frametransformer: 3 timed steps each) that the default single-GPU run appends under \"secondary\"")
    ap.add_argument("--bucket-mb", type=float, default=13.0,
                    help="gradient bucket size in MiB: about one transformer layer (12.6 MB fp32 at d = 512), so that a layer's "
                    "all-reduce starts when that layer's backward ends -- with 32 MiB buckets 48 MB of gradients became ready in "
                    "the last 0.15 ms of backward, with 13 MiB 27 MB (gradient_exchange.bucket_timeline)")
    ap.add_argument("--grad-dtype", choices=["fp32", "bf16", "fp16"], default="fp32", help="element type of the gradient buckets on "
                    "the wire (bf16: 57.7 MB instead of 115 MB per step for the d=512 model; the sum stays fp32 on either side)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of "
                    "replaying one captured hipGraph per step")
    ap.add_argument("--workload", choices=["vivit", "pyramid", "crossmodal", "longclip", "frametransformer"], default="vivit",
                    help="vivit = the metric workload (default); pyramid = BASELINE configs[2] (ResNet-18 3-scale "
                    "pyramid front-end -> the same transformer); crossmodal = configs[3] (+ 32 audio tokens, "
                    "cross-attention block, distillation head); longclip = configs[4] (T=64, 288^2, fp16 + dynamic loss "
                    "scaling, activation checkpointing; reports HBM GB/s of the streaming kernels and the activation "
                    "peak); frametransformer = the reference's default FrameTransformer(model='vid'): R(2+1)D-18 on 14 chunks "
                    "of 12 x 112^2 frames per sample, post-norm encoder with dropout 0.5, 2 samples per GPU "
                    "(config.yaml:2).  Secondary lines, same JSON contract.")
    ap.add_argument("--strong", action="store_true", help="strong scaling (SURVEY 8d secondary metric): the global batch stays "
                    "--batch and each rank takes batch / world clips")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16", help="kernel element type of the vivit workload "
                    "(fp16 adds the device-side dynamic loss scaling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend used for bootstrap, barriers and the timing "
                    "reduction (nccl = RCCL; gloo lets several ranks share one GPU to rehearse the data-parallel path on a "
                    "single-GPU box -- the gradient exchange then goes through torch.distributed as well)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and the RCCL communicator even "
                    "with one rank (rehearses the multi-GPU code path on a single GPU)")
    ap.add_argument("--no-cls-fold", action="store_true", help="A/B switch: run the last space layer's single-query attention "
                    "without folding the K / V projections into the query (functional.CLS_FOLD_MIN_ROWS)")
    ap.add_argument("--no-pair-launch", action="store_true", help="A/B switch: weight and data gradient of the launch-bound "
                    "Linears as two launches (ops.PAIR_LAUNCH)")
    ap.add_argument("--attn-two-pass", action="store_true", help="A/B switch: attention backward as the dq + dk/dv kernel pair "
                    "instead of the one-pass kernel (ops.ATTN_BWD_TWO_PASS)")
    ap.add_argument("--detail-out", default=DETAIL_DEFAULT, help="file (relative to the repository root) that receives the FULL "
                    "record -- per-family rooflines, per-kernel HBM rates, bucket timeline; the printed line is the compact form "
                    "and names this file; '' writes none")
    ap.add_argument("--rendezvous-only", action="store_true", help="launcher self-test (runs without a GPU): every rank joins "
                    "the process group, sums its rank over the group and rank 0 prints a JSON line with n_gpus = world")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])            # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong rank count")
    if args.rendezvous_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank} pid {os.getpid()}")
        if rank == 0:
            print(json.dumps({"metric": "launcher self-test", "n_gpus": world, "rank_sum": float(t), "ranks": names}), flush=True)
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    use_dist = world > 1 or args.force_dist
    comm = None
    if use_dist:
        # A collective that some rank never joins hangs every rank silently: a watchdog turns that into a loud exit (the
        # launcher / torchrun then ends the other ranks) instead of a run that never returns its line.
        _WATCHDOG.update(limit=float(os.environ.get("DVT_BENCH_TIMEOUT_S", "900")), rank=rank)
        progress("process group")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.strong:
        if args.batch % world:
            raise SystemExit(f"--strong: global batch {args.batch} is not divisible by {world} ranks")
        args.batch //= world

    import dvt_amd  # noqa: F401
    if args.no_cls_fold:
        dvt_amd.functional.CLS_FOLD_MIN_ROWS = 1 << 62
    if args.no_pair_launch:
        dvt_amd.ops.PAIR_LAUNCH = False
    if args.attn_two_pass:
        dvt_amd.ops.ATTN_BWD_TWO_PASS = True
    if use_dist and args.backend == "nccl":
        from dvt_amd.dp import Communicator
        try:
            comm = Communicator.from_torch_distributed()
        except Exception as e:
            # No fallback: a run that exchanged its gradients through torch.distributed's eager collectives would measure a
            # different design under the same metric name.  Fail loudly; the launcher ends the other ranks.
            sys.stderr.write(f"[bench] rank {rank}: dvt_comm_init failed under --backend nccl ({type(e).__name__}: {e}); "
                             "refusing to fall back to torch.distributed collectives\n")
            sys.stderr.flush()
            try:
                dist.destroy_process_group()
            finally:
                raise SystemExit(3)
    _LIVE["comm"] = comm

    out = run_workload(args, args.workload, rank, world, use_dist, comm, steps=args.steps, warmup=args.warmup,
                       roofline=not args.no_roofline)
    if world == 1 and not use_dist and args.workload == "vivit" and not args.no_secondary:
        # BASELINE configs[2..4] -- the per-frame CNN encoder + pyramid, the cross-modal attention + distillation head, the
        # long-clip stress -- and the reference's default FrameTransformer(model='vid') (R(2+1)D-18 encoder), timed in the same
        # driver-visible line, bounded (3 timed steps each; the two CNN workloads carry their conv / BatchNorm roofline)
        sec = {}
        for wl in ("pyramid", "crossmodal", "longclip", "frametransformer"):
            release_gpu_memory()
            try:
                r = run_workload(args, wl, rank, world, False, None, steps=3, warmup=1,
                                 roofline=wl in ("pyramid", "frametransformer", "longclip") and not args.no_roofline,
                                 secondary=True)
                sec[wl] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "launch", "peak_hbm_GiB",
                                             "final_loss", "roofline") if k in r}
                sec[wl]["workload"] = r["config"]["workload"]
            except Exception as e:   # a secondary line must never take the headline down with it
                sec[wl] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if out is not None:
            out["secondary"] = sec
    progress(None)                      # done: nothing collective left that could hang
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.workload == "vivit":
            out["cpu_baseline"] = cpu_baseline(dict(image=224, patch=16, classes=19, T=32, d=512, depth=4, heads=8, dh=64))
        emit(out, args.detail_out)
    if _LIVE["comm"] is not None:
        _LIVE["comm"].destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
