"""bench.py --gpus N without a launcher starts N ranks itself (the parent never touches the GPU); here with the
CPU rehearsal switch: the ranks rendezvous over gloo, rank 0 prints one JSON line that says n_gpus = N."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=600)


def test_bench_starts_its_own_ranks():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"DTL_BENCH_REHEARSAL_GLOO": "cpu"})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rank_sum"] == 3.0
    assert "2 ranks joined" in out.stderr


def test_bench_rejects_a_mismatched_world_size():
    out = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0
    assert "--gpus 4 but WORLD_SIZE=2" in (out.stderr + out.stdout)


def test_bench_starts_eight_ranks():
    """configs[3]'s process count: eight ranks rendezvous on 127.0.0.1 and every one reports the world size it joined."""
    out = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], {"DTL_BENCH_REHEARSAL_GLOO": "cpu", "OMP_NUM_THREADS": "1"})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["rank_sum"] == 36.0


def test_a_hung_exchange_ends_the_run_with_a_failure_status():
    """The N > 1 exchange sits behind a watchdog: when it does not come back, rank 0 prints the ONE line it has (with the
    failure named in extra.root_fed) and the run exits non-zero -- never rc 0."""
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"],
               {"DTL_BENCH_REHEARSAL_GLOO": "cpu", "DTL_BENCH_REHEARSAL_HANG": "1", "DTL_BENCH_ROOT_FED_TIMEOUT": "3"})
    assert out.returncode != 0, out.stdout
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert "did not finish" in rec["extra"]["root_fed"]["error"]
    assert "exiting with status 3" in out.stderr
