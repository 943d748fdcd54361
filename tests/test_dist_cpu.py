"""N > 1 host path on CPU: sharding, launcher, unique-id rendezvous, and a
world_size-2 and -3 gloo jobs through the product's gather / merge code."""
import json
import os
import socket
import subprocess
import sys
import textwrap
import threading
import time

import pytest

from polyfasta_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_cover_everything_once_and_balance():
    for n in (0, 1, 7, 16):
        for world in (1, 2, 3, 8):
            got = sorted(i for r in range(world) for i in dist.shard_indices(n, r, world))
            assert got == list(range(n))
    # skewed sizes: the greedy split keeps the heaviest rank within one item of the mean
    sizes = [500_000] * 40 + [20_000_000, 9_000_000, 7_000_000]
    shards = dist.shard_by_size(sizes, 4)
    assert sorted(i for s in shards for i in s) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in s) for s in shards]
    assert max(loads) <= 20_000_000          # the 20 MB file alone; round-robin would stack 3 big files on 2 ranks
    assert max(loads[1:]) - min(loads[1:]) <= 500_000
    assert all(s == sorted(s) for s in shards)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3, 8])      # (8: the node north_star names; seven non-root ranks to merge)
def test_two_rank_gloo_gather(tmp_path, world):
    out = tmp_path / "merged.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), "23"]
    env = {**os.environ, "MASTER_ADDR": "127.0.0.1"}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.load(open(out))
    assert res["ok"] and res["world"] == world and res["tmax"] == float(world) and res["n"] > 23


def test_unique_id_rendezvous_file(tmp_path, monkeypatch):
    """rank 0 publishes, the others wait for a complete file; a reader that starts first still gets it"""
    monkeypatch.setenv("PFA_RDZV_FILE", str(tmp_path / "id"))
    blob = bytes(range(128))
    got = {}
    t = threading.Thread(target=lambda: got.setdefault(1, dist.exchange_unique_id(1, None, 128, timeout=20)))
    t.start()
    time.sleep(0.05)
    assert dist.exchange_unique_id(0, lambda: blob, 128) == blob
    t.join()
    assert got[1] == blob
    # launcher-derived name: identical for all ranks of one launcher
    monkeypatch.delenv("PFA_RDZV_FILE")
    monkeypatch.setenv("MASTER_PORT", "29517")
    assert dist.rendezvous_path() == dist.rendezvous_path() and "29517" in dist.rendezvous_path()


def test_spawn_ranks_env_and_failure(tmp_path):
    """the --gpus launcher: N children with RANK / LOCAL_RANK / WORLD_SIZE, a failing rank fails the job,
    and (without the rehearsal switch) asking for more GPUs than are visible exits non-zero"""
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        open(os.path.join(sys.argv[1], "rank" + os.environ["RANK"]), "w").write(
            ",".join(os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")) + "," +
            str(os.path.dirname(os.environ["PFA_RDZV_FILE"]) != ""))
        sys.exit(3 if sys.argv[2] == "fail" and os.environ["RANK"] == "1" else 0)
    """))
    env_before = dict(os.environ)
    os.environ["PFA_ALLOW_SHARED_GPU"] = "1"
    try:
        assert dist.spawn_ranks(3, [sys.executable, str(script), str(tmp_path), "ok"], timeout=120) == 0
        for r in range(3):
            assert open(tmp_path / f"rank{r}").read() == f"{r},{r},3,127.0.0.1,True"
        assert dist.spawn_ranks(2, [sys.executable, str(script), str(tmp_path), "fail"], timeout=120) == 3
    finally:
        os.environ.clear()
        os.environ.update(env_before)
    if (dist.visible_gpu_count() or 0) < 64:     # there is no box with 64 GPUs: the launcher must refuse
        have = dist.visible_gpu_count()
        if have is not None:
            assert dist.spawn_ranks(64, [sys.executable, str(script), str(tmp_path), "ok"], timeout=120) == 2


def test_cli_gpus_flag_refuses_a_mismatched_launcher(monkeypatch, capsys):
    from polyfasta_amd import cli
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    assert cli.main(["-f", "x.fa", "--gpus", "4"]) == 2
    assert "--gpus 4" in capsys.readouterr().err


def test_unique_id_rendezvous_under_torchrun(tmp_path):
    """the launcher-derived rendezvous name under the real `torch.distributed.run` (how the driver starts
    bench.py on N GPUs): both ranks end up with the bytes rank 0 made, and the file is a fresh one"""
    script = tmp_path / "rdzv.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        from polyfasta_amd import dist
        rank, local, world = dist.env_world()
        assert "PFA_RDZV_FILE" not in os.environ
        blob = dist.exchange_unique_id(rank, lambda: os.urandom(128), 128, timeout=60)
        open(os.path.join({str(tmp_path)!r}, f"got{{rank}}"), "wb").write(blob)
        open(os.path.join({str(tmp_path)!r}, f"path{{rank}}"), "w").write(dist.rendezvous_path())
    """))
    for attempt in range(2):          # a second job of the same shape must not pick up the first one's id
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
        env = {k: v for k, v in os.environ.items() if k != "PFA_RDZV_FILE"}
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        a, b = open(tmp_path / "got0", "rb").read(), open(tmp_path / "got1", "rb").read()
        assert a == b and len(a) == 128
        assert open(tmp_path / "path0").read() == open(tmp_path / "path1").read()
        if attempt:
            assert a != first
        first = a


def test_work_queue_hands_out_every_item_once(tmp_path):
    """the shared --dir work queue: four processes claim concurrently; every item exactly once, largest first"""
    script = tmp_path / "claim.py"
    script.write_text(textwrap.dedent(f"""
        import json, sys, time
        sys.path.insert(0, {ROOT!r})
        from polyfasta_amd import dist
        sizes = [1000 + (i * 7919) % 997 for i in range(1003)]
        q = dist.WorkQueue(sizes, 4, {str(tmp_path / "q")!r})
        got = []
        while True:
            c = q.claim()
            if not c:
                break
            got.append(c)
            time.sleep(0.05)      # a claim is followed by work: the other ranks get their turn
        q.close()
        json.dump(got, open({str(tmp_path)!r} + "/got" + sys.argv[1], "w"))
    """))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)]) for r in range(4)]
    assert all(p.wait(timeout=300) == 0 for p in procs)
    claims = [c for r in range(4) for c in json.load(open(tmp_path / f"got{r}"))]
    items = [i for c in claims for i in c]
    assert sorted(items) == list(range(1003))
    sizes = [1000 + (i * 7919) % 997 for i in range(1003)]
    assert all(len(json.load(open(tmp_path / f"got{r}"))) > 0 for r in range(4))       # nobody starved
    first = min(claims, key=lambda c: -sizes[c[0]])
    assert sizes[first[0]] == max(sizes)                                                # the largest files go first


def test_bench_and_cli_gpus_flag_never_fall_back_silently(tmp_path):
    """`--gpus 2` where the GPUs are missing (this container has none): non-zero exit and no JSON line / no
    rows on stdout — never a silent run on fewer GPUs"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and '"n_gpus"' not in p.stdout, (p.returncode, p.stdout[-300:])
    fa = tmp_path / "a.fa"
    fa.write_text(">x\\nACGT\\n>y\\nACGA\\n")
    p = subprocess.run([sys.executable, "-m", "polyfasta_amd", "-f", str(fa), "--gpus", "2"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "a.fa," not in p.stdout, (p.returncode, p.stdout[-300:])


def test_cli_gpus_flag_starts_the_ranks_with_the_same_arguments(monkeypatch):
    from polyfasta_amd import cli, dist
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(dist, "spawn_ranks", lambda n, argv, timeout=None: seen.update(n=n, argv=argv) or 0)
    assert cli.main(["-d", "somedir", "--cds", "--gpus", "8", "-o", "out.csv"]) == 0
    assert seen["n"] == 8 and seen["argv"][1:3] == ["-m", "polyfasta_amd"]
    assert seen["argv"][3:] == ["-d", "somedir", "--cds", "--gpus", "8", "-o", "out.csv"]


def test_work_queue_by_count_and_stale_counter_file(tmp_path, monkeypatch):
    """A count instead of sizes: claims follow the item order (identical on every rank without a stat()
    per file); a counter file older than the launcher is emptied instead of trusted."""
    qpath = str(tmp_path / "q")
    q = dist.WorkQueue(10, 2, qpath)
    assert q.grain == 1 and [q.claim() for _ in range(3)] == [[0], [1], [2]]
    q.close()
    # the same file, as a job that died left it: 3 items claimed, written long before this launcher started
    os.utime(qpath, (1_000_000, 1_000_000))
    q2 = dist.WorkQueue(10, 2, qpath)
    assert q2.claim() == [0]
    q2.close(unlink=True)
    assert not os.path.exists(qpath)


def test_rank_binds_to_the_cpus_next_to_its_gpu(tmp_path, monkeypatch):
    """dist.gpu_local_cpus / bind_to_gpu_node from a fake sysfs: kfd nodes (two CPU nodes, four GPUs on two
    PCIe roots), the PCI address out of domain / location_id, local_cpulist of that address; ranks on the
    same NUMA node split its CPUs; a missing tree binds nothing."""
    kfd, pci = tmp_path / "kfd", tmp_path / "pci"
    gpus = [(0, 0x0500), (0, 0x2600), (1, 0x8500), (1, 0xa600)]
    for i in range(2):      # CPU-only nodes come first in the kfd topology
        (kfd / str(i)).mkdir(parents=True)
        (kfd / str(i) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\n")
    allowed = sorted(os.sched_getaffinity(0))
    lists = [allowed[:max(1, len(allowed) // 2)], allowed[len(allowed) // 2:] or allowed]
    drm = tmp_path / "drm"
    for g, (domain, loc) in enumerate(gpus):
        (kfd / str(2 + g)).mkdir()
        # GPUs 0, 1 are found through their DRM render node, GPUs 2, 3 through their PCI address
        minor = 128 + 8 * g if g < 2 else 0
        (kfd / str(2 + g) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndomain {domain}\nlocation_id {loc}\ndrm_render_minor {minor}\n")
        addr = f"{domain:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7:x}"
        where = (drm / f"renderD{minor}" / "device") if minor else (pci / addr)
        where.mkdir(parents=True)
        cl = lists[g // 2]
        (where / "local_cpulist").write_text(f"{cl[0]}-{cl[-1]}\n" if cl == list(range(cl[0], cl[-1] + 1)) else ",".join(map(str, cl)) + "\n")
    (kfd / "9").mkdir()                                    # another tenant's GPU: its properties cannot be read
    (kfd / "9" / "properties").mkdir()
    roots = dict(kfd_root=str(kfd), pci_root=str(pci), drm_root=str(drm))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "PFA_NO_AFFINITY"):
        monkeypatch.delenv(var, raising=False)
    assert dist.gpu_local_cpus(0, **roots) == lists[0] and dist.gpu_local_cpus(3, **roots) == lists[1]
    assert dist.gpu_local_cpus(4, **roots) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3,0")
    assert dist.gpu_local_cpus(0, **roots) == lists[1] and dist.gpu_local_cpus(1, **roots) == lists[0]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    before = os.sched_getaffinity(0)
    try:
        got = dist.bind_to_gpu_node(1, 4, **roots)       # GPUs 0 and 1 share node 0: rank 1 takes its second half
        share = lists[0][len(lists[0]) // 2:] or lists[0]
        assert got == share and sorted(os.sched_getaffinity(0)) == share
        assert os.environ.get("PFA_CPUS_ARE_MINE") == "1"      # the loader pool is sized from this share, not split again
    finally:
        os.sched_setaffinity(0, before)
        os.environ.pop("PFA_CPUS_ARE_MINE", None)
    assert dist.bind_to_gpu_node(0, 1, kfd_root=str(tmp_path / "nope"), pci_root=str(pci)) is None
    assert os.sched_getaffinity(0) == before
    monkeypatch.setenv("PFA_NO_AFFINITY", "1")
    assert dist.bind_to_gpu_node(0, 4, **roots) is None


def test_multi_node_launch_is_refused(monkeypatch):
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert dist.single_node(4) and not dist.single_node(8)
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert dist.single_node(8)


def test_ranks_of_one_node_share_its_cpus_by_physical_core(tmp_path):
    """A NUMA node lists its CPUs as 0-3,8-11 (8-11 the sibling threads of 0-3): two ranks get two whole cores each, not the
    first threads of all four cores and their siblings (dist.by_core)."""
    root = tmp_path / "cpu"
    for c in range(4):
        for t in (c, c + 8):
            (root / f"cpu{t}" / "topology").mkdir(parents=True)
            (root / f"cpu{t}" / "topology" / "thread_siblings_list").write_text(f"{c},{c + 8}\n")
    cpus = [0, 1, 2, 3, 8, 9, 10, 11]
    order = dist.by_core(cpus, str(root))
    assert order == [0, 8, 1, 9, 2, 10, 3, 11]
    assert sorted(order[:4]) == [0, 1, 8, 9] and sorted(order[4:]) == [2, 3, 10, 11]
    assert dist.by_core(cpus, str(tmp_path / "nope")) == cpus            # no topology: the list as it is
