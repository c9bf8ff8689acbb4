"""bench.py --gpus N without a launcher (the reference is pl.Trainer(gpus=1), src/main.py:87): the parent
spawns N rank processes before any GPU call, relays rank 0's JSON line, refuses a world size that differs from --gpus and
exits non-zero when a rank fails.  Runs on CPU (gloo) through the launcher's rendezvous-only mode."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=180):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=e)


def test_gpus_2_spawns_two_ranks_and_reports_n_gpus_2():
    r = _run(["--gpus", "2", "--backend", "gloo", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0 and len(out["ranks"]) == 2
    assert len({s.split("pid")[1] for s in out["ranks"]}) == 2          # two different processes


def test_world_size_that_differs_from_gpus_is_refused():
    r = _run(["--gpus", "2", "--rendezvous-only"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr
    r = _run(["--gpus", "1", "--backend", "gloo", "--rendezvous-only"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr


def test_failing_rank_fails_the_launch():
    # an unknown backend makes every child raise during init_process_group: the parent must not print a line
    r = _run(["--gpus", "2", "--backend", "no_such_backend", "--rendezvous-only"])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_printed_line_is_compact_and_carries_every_workload(tmp_path, capsys):
    """bench.py prints ONE line that fits the driver's 8 KB tail whatever the full record holds: every
    workload's value / ms_per_step, the dominant family's roofline, the long clip's HBM report and the CPU baseline stay in
    the line; the per-family detail goes to the file the line names."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fams = {f"family {i} " + "x" * 120: {"ms_per_step": 1.234, "launches_per_step": 12, "achieved": 512.3, "frac": 0.2049 - i * 0.01,
                                         "algorithmic_MB_per_launch": 123.4, "traffic_ratio": 1.0 + 0.2 * i} for i in range(12)}
    cnn = {"bound": "mfma", "peak": 2500.0, "unit": "TFLOP/s", "conv_families": fams, "traffic_source": "profiles/x.json",
           "hbm_kernels": {f"bn_{k}": {"ms_per_step": 0.5, "launches_per_step": 30, "algorithmic_GBps": 4000.0, "frac_of_hbm_peak": 0.5}
                           for k in ("stats", "apply_fwd", "bwd")}, "achieved": 512.3, "frac": 0.2049, "kernel": "family 0 " + "x" * 120}
    lc = {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "kernel": "layernorm_bwd", "achieved": 5600, "frac": 0.7,
          "peak_activation_GiB": 6.3, "traffic": None,
          "kernels": {k: {"us": 123.4, "n": 12, "GBps": 4321, "frac": 0.54, "mfma_frac": 0.123}
                      for k in ("layernorm_fwd", "layernorm_bwd", "attention_fwd", "attention_bwd", "patchify", "tokens_assemble_fwd",
                                "ff1_gemm_gelu_epilogue", "attn_cls_fwd", "attn_cls_bwd")}}
    sec = {wl: {"metric": "m" * 100, "value": 123.45, "unit": "clips/s", "ms_per_step": 12.345, "dtype": "bf16", "launch": "hipGraph replay",
                "peak_hbm_GiB": 6.5, "final_loss": 0.4, "workload": "w" * 300, "roofline": r}
           for wl, r in (("pyramid", cnn), ("crossmodal", None), ("longclip", lc), ("frametransformer", cnn))}
    out = {"metric": "clips/sec fwd+bwd, B=8 T=32 3x224x224 bf16", "value": 1400.0, "unit": "clips/s", "n_gpus": 8, "steps": 20, "warmup": 5,
           "ms_per_step": 5.7, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "W" * 400, "global_batch": 64, "parallelism": "dp8", "params_M": 28.84},
           "launch": "hipGraph replay (bucketed RCCL all-reduce captured inside the step graph)", "final_loss": 0.4,
           "step_ms": {"p10": 5.6, "median": 5.7, "p90": 5.8}, "optimizer_ms_per_step": 0.16, "peak_hbm_GiB": 3.8,
           "executed_mfma_frac": 0.22, "model_mfma_frac": 0.29,
           "roofline": {"bound": "mfma", "kernel": "gemm_dma_kernel<A k-major, B k-major> (forward Linear)", "achieved": 700.0, "peak": 2500.0,
                        "unit": "TFLOP/s", "frac": 0.28, "traffic": 308787892, "traffic_source": "t" * 200, "algorithmic_bytes_per_launch": 273577511,
                        "traffic_ratio": 1.129, "launches": 13, "avg_launch_us": 107.7, "families": {f"f{i}" * 30: {"ms_per_step": 1.0} for i in range(3)},
                        "hbm_kernels": {f"k{i}": {"us_per_launch": 1.0, "note": "n" * 300} for i in range(12)},
                        "same_kernel_full_k": {"frac": 0.47}, "timing_note": "z" * 300},
           "gradient_exchange": {"world": 8, "rank_devices": [f"rank {r}: cuda:{r} AMD Instinct MI355X pid {1000 + r}" for r in range(8)],
                                 "dtype": "fp32", "bucket_mb": 13.0, "through": "dvt_comm_allreduce (RCCL behind the C ABI)",
                                 "eager_ms_per_step": 6.0, "graph_ms_per_step": 5.9, "graph_ms_per_step_without_exchange": 5.6,
                                 "exposed_allreduce_ms_per_step": 0.3, "allreduce_bytes_per_step": 115368192,
                                 "bucket_timeline": [{"wire_bytes": 16e6, "start_ms_after_backward_end": -2.0, "end_ms_after_backward_end": -1.5, "ms": 0.5}] * 10},
           "secondary": sec,
           "cpu_baseline": {"value": 0.5, "unit": "clips/s", "cores": 16, "kind": "port", "sample": "s" * 300, "host_logical_cpus": 256}}
    detail = os.path.relpath(str(tmp_path / "detail.json"), ROOT)
    txt = bench.emit(out, detail)
    assert len(txt) < bench.LINE_LIMIT < 8192 and "\n" not in txt
    line = json.loads(txt)
    assert capsys.readouterr().out.strip() == txt
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "secondary", "detail"):
        assert k in line, k
    assert set(line["secondary"]) == {"pyramid", "crossmodal", "longclip", "frametransformer"}
    for wl, s in line["secondary"].items():
        assert s["value"] == 123.45 and s["ms_per_step"] == 12.345, wl
    assert line["roofline"]["frac"] == 0.28 and line["roofline"]["traffic"] == 308787892
    assert line["secondary"]["longclip"]["roofline"]["kernels"]["layernorm_fwd"]["GBps"] == 4321
    assert line["secondary"]["pyramid"]["roofline"]["conv_frac_min"] == min(f["frac"] for f in fams.values())
    assert line["gradient_exchange"]["exposed_allreduce_ms_per_step"] == 0.3 and line["gradient_exchange"]["devices"]
    with open(tmp_path / "detail.json") as fh:                    # the full record is in the file the line names
        assert json.load(fh)["secondary"]["pyramid"]["roofline"]["conv_families"] == fams
