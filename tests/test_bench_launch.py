"""bench.py's own N-rank launch on CPU: `python bench.py --gpus 2 --dry-launch` must start two ranks as child processes
(torch.distributed.run over 127.0.0.1), rendezvous them (gloo), sum-reduce the per-rank tile frames onto rank 0, relay
rank 0's single JSON line and return the children's exit code — the path the driver's multi-GPU run takes when nothing
wraps bench.py in a launcher.  No rendering happens in a dry launch (the hot path has no CPU form)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_starts_its_own_ranks_and_relays_one_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "0"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_launch"] is True and out["reduce_exact"] is True
    assert out["config"]["parallelism"] == "tiles%2" and "reduce" in out["config"]["collective"]
    assert out["steps"] == 2 and out["ms_per_step"] > 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_single_rank_dry_launch_needs_no_process_group():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-launch", "--steps", "1"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["reduce_exact"] is True
