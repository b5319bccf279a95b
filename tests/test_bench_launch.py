"""bench.py's launch contract on CPU: `python bench.py --gpus N` with no launcher around it must re-exec
itself under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) and still print exactly ONE JSON line from rank 0;
under a launcher (WORLD_SIZE set) it must not re-exec.  TANGO_BENCH_STUB=1 swaps the engine for a stand-in compute function and
RCCL for gloo: what runs here is the launch / sharding / gather / JSON plumbing, not a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env=None):
    env = dict(os.environ, TANGO_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    return json.loads(lines[0])


def test_bench_self_launches_for_gpus_2():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "0", "--batch", "3"])
    assert out["stub"] is True and out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 2
    assert out["config"]["global_batch"] == 6          # weak scaling: 3 prompts per rank
    # the line shows a straggler rank or a slow collective, not one number
    pr = out["per_rank_ms"]
    assert len(pr["all"]) == 2 and 0.0 <= pr["min"] <= pr["max"] and out["bcast_ms"] >= 0.0 and out["gather_ms"] >= 0.0


def test_bench_single_process_default():
    out = _run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--batch", "2"])
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == 2


def test_bench_under_an_external_launcher_does_not_relaunch():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1"])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2
