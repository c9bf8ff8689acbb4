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


def cpu_baseline(cfg, seconds_budget=25.0):
    """Oracle fwd+bwd on host cores, bounded sample of the same workload."""
    from oracle import clip_path as O
    torch.manual_seed(1130)
    from dvt_amd.models.vit import ViViT
    net = ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["T"], dim=cfg["d"], depth=cfg["depth"],
                heads=cfg["heads"], dim_head=cfg["dh"])
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    bs = 1
    x = torch.randn(bs, cfg["T"], 3, cfg["image"], cfg["image"])
    y = (torch.rand(bs, cfg["classes"]) < 0.2).float()
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    O.vivit_step_fwd_bwd(x, y, P, patch=cfg["patch"], depth=cfg["depth"], heads=cfg["heads"])   # warm-up
    warm = time.perf_counter() - t0
    steps = max(1, min(5, int(seconds_budget / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(steps):
        O.vivit_step_fwd_bwd(x, y, P, patch=cfg["patch"], depth=cfg["depth"], heads=cfg["heads"])
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(bs / dt, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"oracle (pure-torch fp32 restatement) fwd+bwd on {bs} clip of the metric shape "
                      f"(T={cfg['T']}, {cfg['image']}^2, d={cfg['d']}), 1 warm-up + {steps} timed steps, "
                      f"{dt:.2f} s/step"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of "
                    "replaying one captured hipGraph per step (single-GPU only)")
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
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets several ranks share "
                    "one GPU to rehearse the data-parallel path on a single-GPU box)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even with one "
                    "rank (rehearses the multi-GPU code path on a single GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.strong:
        if args.batch % world:
            raise SystemExit(f"--strong: global batch {args.batch} is not divisible by {world} ranks")
        args.batch //= world

    import dvt_amd
    from dvt_amd import functional as F
    from dvt_amd import ops
    from dvt_amd.dp import FlatParameters
    from dvt_amd.models.vit import ViViT

    cfg = dict(image=224, patch=16, classes=19, T=32, d=512, depth=4, heads=8, dh=64)
    cdt = torch.bfloat16
    if args.workload == "longclip":
        cfg.update(image=288, T=64)
        cdt = torch.float16
    elif args.dtype == "fp16":
        cdt = torch.float16
    torch.manual_seed(1130)                                       # src/main.py:25
    if args.workload == "frametransformer":
        from dvt_amd.models.frame_transformer import FrameTransformer
        if args.batch == 8:
            args.batch = 2                                     # config.yaml:2
        net = FrameTransformer(batch_size=args.batch, seq_len=13, cls=1, model="vid", opt="adamW", learning_rate=5e-6,
                               weight_decay=0.09, momentum=0.005).cuda().train()
    elif args.workload in ("vivit", "longclip"):
        net = ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["T"], dim=cfg["d"], depth=cfg["depth"],
                    heads=cfg["heads"], dim_head=cfg["dh"], compute_dtype=cdt,
                    activation_checkpointing=args.workload == "longclip").cuda().train()
    else:
        from dvt_amd.models.pyramid_vivit import PyramidViViT
        cm = args.workload == "crossmodal"
        net = PyramidViViT(cfg["image"], cfg["classes"], cfg["T"], dim=cfg["d"], depth=cfg["depth"], heads=cfg["heads"],
                           dim_head=cfg["dh"], audio_tokens=32 if cm else 0, audio_dim=128, distill=cm,
                           compute_dtype=torch.bfloat16).cuda().train()
    flat = FlatParameters(net, bucket_mb=args.bucket_mb, compute_dtype=cdt)
    flat.broadcast_parameters(0)
    flat.sync_compute_copy()

    gen = torch.Generator().manual_seed(1130 + rank)
    B = args.batch
    if args.workload == "frametransformer":
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
    if args.workload == "crossmodal":       # one 128-d VGGish-style vector per 1-s chunk (SURVEY 8d synthetic inputs)
        audio = torch.randn(B, 32, 128, generator=gen).to(torch.bfloat16).cuda()

    def step():
        flat.zero_grad()
        if args.workload in ("vivit", "longclip"):
            loss = F.bce_with_logits(net(x), y)
        elif args.workload == "frametransformer":
            loss = net.training_step((y, None, x), 0)
        else:
            loss = net.training_step((y, x, audio) if audio is not None else (y, x))
        loss.backward(gloss)
        flat.finish_backward()
        flat.adamw_step(lr=5e-6, weight_decay=0.09)              # config.yaml:3,12
        return loss

    def barrier():
        if use_dist:
            dist.barrier()

    # Single GPU: the whole step (fwd + loss + bwd + AdamW + weight cast) is captured once as a
    # hipGraph and replayed.  Multi GPU: eager launches, so that the bucketed RCCL all-reduces
    # stay ordinary asynchronous collectives overlapped with backward.
    use_graph = not use_dist and not args.no_graph
    run = step
    if use_graph:
        from dvt_amd.graph import capture_step
        replay, static_loss = capture_step(step, warmup=2)

        def run():
            replay()
            return static_loss

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step spread (p10 / median / p90)
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        loss = run()
        evs[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.detach())
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
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
    if not args.no_roofline:
        prof = EventProfiler()
        ops.set_profiler(prof)
        nprof = max(1, min(3, args.steps))
        # The host needs longer to issue a step than the GPU to run it when every launch is bracketed by events; a
        # device-side delay in front of each profiled step lets the host run ahead, so the brackets see device time only.
        host_ms = max(20.0, 4.0 * elapsed / args.steps * 1e3)
        ops.device_delay(20000)
        bracket_us = prof.calibrate(lambda: ops.device_delay(0)) * 1e3
        for _ in range(nprof):
            for _ in range(int(host_ms // 200) + 1):
                ops.device_delay(int(min(host_ms, 200.0) * 1000))
            step()
        torch.cuda.synchronize()
        ops.set_profiler(None)
        summ = prof.summary()
        fam = {}
        hbm = {}
        big = {}
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
        for key, (ms, fl, cnt) in summ.items():
            if key[0] != "gemm":
                continue
            tag, ak, bk, M, N, K = key
            if M * N * K < (1 << 30):      # launch-bound temporal/head GEMMs: not this kernel's regime
                continue
            f = fam.setdefault((ak, bk), [0.0, 0.0, 0])
            f[0] += ms; f[1] += fl; f[2] += cnt
        names = {(1, 1): "gemm_dma_kernel<A k-major, B k-major> (forward Linear)",
                 (1, 0): "gemm_dma_kernel<A k-major, B mn-major> (data gradient)",
                 (0, 0): "gemm_dma_kernel<A mn-major, B mn-major> (weight gradient, split-K incl. reduce)"}
        pmc = {}
        try:   # per-launch HBM bytes of each family from the committed PMC profile (rocprofv3 --pmc)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
                pmc = json.load(fh)["families"]
        except Exception:
            pmc = {}
        pmc_key = {(1, 1): "fwd", (1, 0): "dgrad", (0, 0): "wgrad"}
        if fam:
            dom = max(fam, key=lambda k: fam[k][0])
            ms, fl, cnt = fam[dom]
            ach = fl / (ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": names.get(dom, str(dom)), "achieved": round(ach, 1),
                    "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                    "traffic": (pmc.get(pmc_key.get(dom, ""), {}) or {}).get("hbm_bytes_corrected"),
                    "traffic_note": "bytes/launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE), profiles/r01_pmc_traffic.json; "
                                    "FETCH_SIZE includes Infinity-Cache hits (upper bound of HBM reads)",
                    "algorithmic_bytes_per_launch": None, "launches": cnt // nprof, "avg_launch_us": round(ms * 1e3 / cnt, 1),
                    "families": {names.get(k, str(k)): {"ms_per_step": round(v[0] / nprof, 3),
                                                        "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1)}
                                 for k, v in fam.items()},
                    "hbm_kernels": hbm,
                    "event_bracket_overhead_us": round(bracket_us, 1),
                    "timing_note": "HIP-event pair per launch on the launch stream, behind a device-side delay so that the host "
                                   "runs ahead; the measured cost of an empty bracket is subtracted from every record"}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        clips = B * world * args.steps / elapsed
        n_tok = (cfg["image"] // cfg["patch"]) ** 2 + 1
        fwd, tot = algorithmic_flops_per_clip(cfg["T"], n_tok, cfg["d"], cfg["heads"], cfg["dh"], cfg["depth"],
                                              3 * cfg["patch"] ** 2, n_tok - 1)
        out = {
            "metric": f"clips/sec fwd+bwd, B=8 T=32 3x224x224 {args.dtype}" if args.workload == "vivit" else
                      ("samples/sec fwd+bwd [frametransformer workload], B=2 x 14 chunks x 12 x 3x112x112"
                       if args.workload == "frametransformer" else
                       f"clips/sec fwd+bwd [{args.workload} workload], B={B} T={cfg['T']} 3x{cfg['image']}x{cfg['image']}"),
            "value": round(clips, 2), "unit": "samples/s" if args.workload == "frametransformer" else "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "fp16" if cdt == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload],
                       "global_batch": B * world, "parallelism": f"dp{world}", "params_M": round(flat.total / 1e6, 2)},
            "launch": "hipGraph replay" if use_graph else "eager",
            "model_tflops": round(tot * B * world / (elapsed / args.steps) / 1e12, 1)
            if args.workload in ("vivit", "longclip") else None,
            "model_mfma_frac": round(tot * B / (elapsed / args.steps) / 1e12 / MFMA_PEAK_TFLOPS, 4)
            if args.workload in ("vivit", "longclip") else None,
            "final_loss": round(final_loss, 5),
            "step_ms": {"p10": round(pct(0.1), 3), "median": round(pct(0.5), 3), "p90": round(pct(0.9), 3)},
            "optimizer_ms_per_step": round(optimizer_ms, 3),
            "peak_hbm_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
        }
        if cdt == torch.float16:
            out["loss_scale"] = float(flat.scale_dev)
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline and args.workload == "vivit":
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
