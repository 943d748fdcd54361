"""N > 1 host path on CPU: world_size-2 gloo job."""
import json
import os
import socket
import subprocess
import sys

from polyfasta_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 7, 16):
        for world in (1, 2, 3, 8):
            got = sorted(i for r in range(world) for i in dist.shard_indices(n, r, world))
            assert got == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_gather(tmp_path):
    out = tmp_path / "merged.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), "23"]
    env = {**os.environ, "MASTER_ADDR": "127.0.0.1"}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.load(open(out))
    assert res["ok"] and res["world"] == 2 and res["tmax"] == 2.0 and res["n"] > 23
