"""CPU: bench.py's N>1 code path itself (2 ranks, gloo, --dry-run) so that the first real multi-GPU run is not the first execution of
the sharding / gather / max-over-ranks / JSON code.  Both launch forms: the documented `python -m torch.distributed.run ... bench.py
--gpus N` and the plain `python bench.py --gpus N` the driver's N=1 record shows (bench.py then starts its own N ranks)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_dry_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "3",
           "--max-boxes", "512", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                         # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["dry_run"] is True
    assert j["gather_ok"] is True and j["config"]["global_batch"] == 6 and j["value"] > 0
    # counts first, then only the rows any rank filled: far below the 512-row capacity of the decode block
    assert 0 < j["gather_message_bytes_per_rank"] <= 3 * 400 * 112 * 4 + 3 * 4 + 8
    # the steady-state form: the whole capacity block in ONE collective (tests/test_dist.py asserts it contains no host synchronisation)
    assert j["static_gather_ok"] is True and j["static_gather_message_bytes_per_rank"] == 3 * 512 * 112 * 4


def test_bench_single_rank_dry_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--max-boxes", "512"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["gather_ok"] is True


def test_train_bench_two_ranks_dry_run():
    """bench.py --train --dry-run with 2 ranks under gloo: the bucketed gradient all-reduce (findtextcenternet_amd.dist.BucketedAllReduce),
    max-over-ranks timing and the JSON line of the train bench."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--train", "--dry-run", "--batch", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["allreduce_ok"] is True and j["buckets"] == 3 and j["config"]["global_batch"] == 16


def test_plain_python_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2 --dry-run` with no launcher and no WORLD_SIZE: bench.py re-executes itself under torch.distributed.run
    (one process per GPU); ONE JSON line, last on stdout, n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "3",
                        "--max-boxes", "512", "--dry-run"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1 and out[-1] == lines[0], r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["gather_ok"] is True and j["static_gather_ok"] is True and j["config"]["global_batch"] == 6


def test_plain_python_train_bench_gpus_2_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--train", "--gpus", "2", "--dry-run", "--batch", "8"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1 and out[-1] == lines[0], r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["allreduce_ok"] is True and j["config"]["global_batch"] == 16


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120,
                       cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
