"""bench.py's rank logic without a GPU: `python bench.py --gpus 2 --dist-selftest` must start two ranks by itself (no launcher), join them through
torch.distributed (gloo here, RCCL on the GPUs), apply the MAX-over-ranks timing rule and gather the per-rank counters; a launcher-provided WORLD_SIZE
that disagrees with --gpus must fail loudly."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_plain_gpus2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dist-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert out["max_elapsed"] == 0.75                        # MAX over ranks of 0.5 + 0.25 * rank
    assert out["per_rank_evals"] == [1000.0, 2000.0] and out["evals"] == 3000.0 and out["frames"] == 16


def test_single_rank_selftest():
    r = _run(["--gpus", "1", "--dist-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["evals"] == 1000.0


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4", "--dist-selftest"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
