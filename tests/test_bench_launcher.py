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
