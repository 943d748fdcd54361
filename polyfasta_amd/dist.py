"""One process per GPU: sharding of independent alignments, the launcher and the final gather.

The `--dir` batch mode of the reference is an embarrassingly parallel loop over files
(PolyFastA.py:104).  Here the ranks of a node drain one work queue of the sorted file list,
largest files first (`WorkQueue`; `shard_by_size` is the static split for callers without a
shared file system view); no collective touches the data path.  When every
rank has scanned its share, the fixed-size integer result records travel through ONE RCCL
all-gather (`pfa_comm_*` of the C ABI, include/pfa.h: librccl over xGMI, bound by the library
itself — no torch anywhere); rank 0 finishes the floats and prints the rows in the
reference's order.

Ranks come from a launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE, e.g.
`python -m torch.distributed.run`), or from `spawn_ranks()` below (`--gpus N` of the CLI and
of bench.py).  The 128-byte RCCL unique id goes from rank 0 to the others through a file.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import pickle
import subprocess
import sys
import tempfile
import time

import numpy as np


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on stdout when the first communicator is built;
    stdout carries the CSV rows (or the bench's one JSON line), so file descriptor 1
    is pointed at stderr while the communicator comes up."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def env_world():
    """(rank, local_rank, world) from the launcher's environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


# ---------------------------------------------------------------- sharding
def shard_by_size(sizes, world: int):
    """Largest-first greedy split of items (file sizes in bytes) over `world` ranks: every item
    goes to the rank with the least bytes so far (ties: lowest rank).  Returns one ascending
    index list per rank; every rank computes the same split from the same sizes."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += max(int(sizes[i]), 1)
    return [sorted(s) for s in shards]


def shard_indices(n_items: int, rank: int, world: int, sizes=None):
    """Indices of the items rank `rank` owns (equal sizes when none are given)."""
    return shard_by_size(sizes if sizes is not None else [1] * n_items, world)[rank]


class WorkQueue:
    """The `--dir` loop as a work queue shared by the ranks of one node (north_star: "an
    embarrassingly-parallel work queue"): the items are claimed a few at a time through a counter
    in a file next to the rendezvous file (`flock`); a rank that finishes early simply claims more,
    so neither skewed file sizes nor a slow GPU decide the job time.  Which rank scanned a file does
    not matter to the output: rank 0 orders the rows by file index.

    `items`: a count — the claim order is the item order, identical on every rank by construction
    (the CLI: the sorted path list, no stat() of 100 k files per rank) — or a list of sizes, claimed
    largest first (callers that know their sizes and that every rank sees the same ones).

    The counter file carries no state of an earlier job: a file older than this job's launcher is
    emptied under the lock before the first claim (the same rule as the unique-id file)."""

    def __init__(self, items, world: int, path=None):
        if isinstance(items, int):
            self.order = list(range(items))
        else:
            self.order = sorted(range(len(items)), key=lambda i: (-int(items[i]), i))
        self.world = max(int(world), 1)
        # a claim = 1/(4 world) of the items at most: even a small directory is spread over the ranks
        self.grain = max(1, min(512, len(self.order) // (4 * self.world)))
        self.path = path or rendezvous_path() + ".queue"
        self.n_claims, self.n_items, self.claim_s = 0, 0, 0.0           # this rank's claims (bench.py: the stage table of the N > 1 leg)
        t0 = time.perf_counter()
        self.fd = os.open(self.path, os.O_RDWR | os.O_CREAT, 0o600)     # empty file = nothing claimed yet
        import fcntl
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        try:
            t_launcher = _parent_start_time()
            if t_launcher and os.fstat(self.fd).st_mtime < t_launcher - 1.0:
                os.ftruncate(self.fd, 0)          # left behind by a job whose rank 0 died before its unlink
        finally:
            fcntl.flock(self.fd, fcntl.LOCK_UN)
        self.open_s = time.perf_counter() - t0

    def claim(self):
        """the next item indices of this rank ([] when the queue is drained)"""
        import fcntl
        t0 = time.perf_counter()
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        try:
            os.lseek(self.fd, 0, os.SEEK_SET)
            raw = os.read(self.fd, 32)
            lo = int(raw) if raw.strip() else 0
            hi = min(lo + self.grain, len(self.order))
            if hi > lo:
                os.lseek(self.fd, 0, os.SEEK_SET)
                os.write(self.fd, b"%-31d" % hi)
        finally:
            fcntl.flock(self.fd, fcntl.LOCK_UN)
        self.n_claims += 1
        self.n_items += hi - lo
        self.claim_s += time.perf_counter() - t0
        return self.order[lo:hi]

    def close(self, unlink=False):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None
        if unlink:
            with contextlib.suppress(OSError):
                os.unlink(self.path)


def single_node(world: int) -> bool:
    """The work queue and the unique-id rendezvous are files of ONE node: a launcher that spreads the
    ranks over several nodes (LOCAL_WORLD_SIZE < WORLD_SIZE) is refused by the CLI and the bench."""
    local = os.environ.get("LOCAL_WORLD_SIZE")
    return local is None or int(local) == int(world)


# ---------------------------------------------------------------- CPU affinity
def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return sorted(cpus)


def gpu_local_cpus(local_rank: int, kfd_root="/sys/class/kfd/kfd/topology/nodes", pci_root="/sys/bus/pci/devices",
                   drm_root="/sys/class/drm", cpu_root=None):
    """CPUs next to GPU `local_rank` (its PCIe root's NUMA node), from sysfs alone — no HIP call, so a
    rank can bind itself before it touches the GPU.  The kfd topology lists the GPUs in HIP's device
    order (nodes this container may not read belong to other tenants and are skipped); a GPU node names
    its DRM render node (`drm_render_minor`) and its PCI address (`domain`, `location_id` = bus << 8 |
    device << 3 | function); `<drm>/renderD<minor>/device/local_cpulist`, else
    `<pci>/<address>/local_cpulist`, holds the CPUs of that device's NUMA node.
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber the devices: the list is followed when it holds
    plain indices.  Returns a sorted CPU list, or None when anything is missing (then nothing is bound)."""
    try:
        gpus = []
        for node in sorted(os.listdir(kfd_root), key=lambda x: int(x)):
            try:
                with open(os.path.join(kfd_root, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v:
                gpus = [gpus[int(x)] for x in v.split(",") if x.strip() != ""]
        props = gpus[local_rank]
        cands = []
        if int(props.get("drm_render_minor", "0")) > 0:
            cands.append(os.path.join(drm_root, f"renderD{int(props['drm_render_minor'])}", "device", "local_cpulist"))
        if "location_id" in props:
            domain, loc = int(props.get("domain", "0")), int(props["location_id"])
            cands.append(os.path.join(pci_root, f"{domain:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7:x}", "local_cpulist"))
        for path in cands:
            try:
                with open(path) as f:
                    cpus = _cpulist(f.read())
                if cpus:
                    return cpus
            except OSError:
                continue
        return None
    except (OSError, ValueError, KeyError, IndexError):
        return None


def by_core(cpus, cpu_root="/sys/devices/system/cpu"):
    """`cpus` ordered so that the hardware threads of one physical core are neighbours and cores keep their order
    (topology/thread_siblings_list); unchanged when sysfs does not say."""
    first = {}
    try:
        for c in cpus:
            with open(os.path.join(cpu_root, f"cpu{c}", "topology", "thread_siblings_list")) as f:
                sib = _cpulist(f.read())
            first[c] = min(sib) if sib else c
    except (OSError, ValueError):
        return list(cpus)
    return sorted(cpus, key=lambda c: (first[c], c))


def bind_to_gpu_node(local_rank: int, local_world: int = 1, **roots):
    """Pin this process (and the loader threads it will start) to the CPUs next to its GPU; ranks
    that share a NUMA node split its CPUs evenly (rank r of the k ranks on that node takes every
    k-th slice).  Called by a rank BEFORE its first GPU call: pinned buffers are then allocated on
    the right node and eight ranks' loader threads do not land on one socket.  PFA_NO_AFFINITY=1
    switches it off.  Returns the CPU list it bound to, or None."""
    if os.environ.get("PFA_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return None
    if os.environ.get("PFA_CPUS_ARE_MINE") and not roots:      # bound already (a process binds once)
        return sorted(os.sched_getaffinity(0))
    mine = gpu_local_cpus(local_rank, **roots)
    if not mine:
        return None
    allowed = sorted(set(mine) & os.sched_getaffinity(0))
    if not allowed:
        return None
    # the ranks whose GPUs sit on the same CPUs share them out — by physical core: a node lists its CPUs as 0-63,128-191 (the
    # second half are the sibling threads of the first), and slices of that list would give two ranks the same cores
    same = [r for r in range(max(local_world, 1)) if gpu_local_cpus(r, **roots) == mine] or [local_rank]
    k, me = len(same), same.index(local_rank) if local_rank in same else 0
    if not os.environ.get("PFA_BIND_BY_CPU_LIST"):             # (timing experiments: slices of the node's CPU list as it stands)
        allowed = by_core(allowed, roots.get("cpu_root", "/sys/devices/system/cpu"))
    share = sorted(allowed[me * len(allowed) // k:(me + 1) * len(allowed) // k] or allowed)
    try:
        os.sched_setaffinity(0, share)
    except OSError:
        return None
    os.environ["PFA_CPUS_ARE_MINE"] = "1"     # the ingest sizes its loader pool from this rank's own CPUs (no further split by ranks)
    return share


# ---------------------------------------------------------------- launcher
def visible_gpu_count():
    """GPUs this process tree can use, WITHOUT touching the HIP runtime (a launcher must not
    initialise the GPU before it starts its ranks): the kfd topology, cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when the topology cannot be read."""
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv, timeout=None) -> int:
    """Start `n` fresh processes `argv` (one per GPU: RANK = LOCAL_RANK = 0..n-1, WORLD_SIZE = n)
    and wait for them.  Called before anything in this process has touched the GPU.  Fails
    loudly (non-zero) when fewer than `n` GPUs are visible: never a silent 1-GPU run.  Returns
    the largest exit code; when a rank fails the others are terminated."""
    have = visible_gpu_count()
    if have is not None and have < n and not os.environ.get("PFA_ALLOW_SHARED_GPU"):
        print(f"polyfasta_amd: --gpus {n} asked for, {have} GPU(s) visible", file=sys.stderr)
        return 2
    rdzv = tempfile.mkdtemp(prefix="pfa_rdzv_")
    port = _free_port()
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL's intra-node transport shares device buffers between the rank
        # processes through IPC handles; the kernel driver of this pool only implements the dmabuf flavour,
        # and with the runtime's legacy default hipIpcGetMemHandle fails ("invalid argument") inside
        # ncclCommInitRank.  The pool's images export it already; a value the user set is kept.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PFA_RDZV_FILE=os.path.join(rdzv, "rccl_id"),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(list(argv), env=env, stdin=None if r == 0 else subprocess.DEVNULL))
    t_end = None if timeout is None else time.time() + timeout
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = max(rc, code if code > 0 else 1)
                    for q in live:          # a collective would wait for the dead rank for ever
                        q.terminate()
            if t_end is not None and time.time() > t_end:
                for q in live:
                    q.terminate()
                rc = max(rc, 124)
                t_end = None
            time.sleep(0.005)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        with contextlib.suppress(OSError):
            for f in os.listdir(rdzv):
                os.unlink(os.path.join(rdzv, f))
            os.rmdir(rdzv)
    return rc


# ---------------------------------------------------------------- unique-id rendezvous
def _parent_start_time():
    """Wall-clock start of the parent process (the launcher), or 0.0."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])       # field 22: starttime
        with open("/proc/stat") as f:
            btime = next(int(line.split()[1]) for line in f if line.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, StopIteration, IndexError):
        return 0.0


def rendezvous_path():
    """Where rank 0 leaves the unique id: PFA_RDZV_FILE (spawn_ranks), else a name all ranks of
    one launcher derive alike — its rendezvous port, its pid (the ranks' parent), its restart count."""
    p = os.environ.get("PFA_RDZV_FILE")
    if p:
        return p
    key = "_".join([os.environ.get("MASTER_PORT", "0"), str(os.getppid()),
                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none")])
    return os.path.join(tempfile.gettempdir(), f"pfa_rccl_id_{os.getuid()}_{key}")


def exchange_unique_id(rank: int, make_id, nbytes: int, timeout: float = 300.0) -> bytes:
    """Rank 0 calls `make_id()` and publishes the bytes (write + rename: readers never see a partial
    file); the other ranks wait for a file that is not older than their launcher."""
    path = rendezvous_path()
    if rank == 0:
        blob = bytes(make_id())
        assert len(blob) == nbytes
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)
        return blob
    t_launcher = _parent_start_time() - 1.0
    t_end = time.time() + timeout
    while True:
        try:
            st = os.stat(path)
            if st.st_size == nbytes and st.st_mtime >= t_launcher:
                with open(path, "rb") as f:
                    blob = f.read()
                if len(blob) == nbytes:
                    return blob
        except OSError:
            pass
        if time.time() > t_end:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout:.0f} s (is rank 0 alive?)")
        time.sleep(0.002)


# ---------------------------------------------------------------- communicator
class Comm:
    """What the CLI and the bench need from a communicator; `RcclComm` is the product's,
    tests inject others through `cli.main(argv, comm=...)`."""
    rank = 0
    world = 1

    def barrier(self):
        raise NotImplementedError

    def max_float(self, x: float) -> float:
        raise NotImplementedError

    def all_gather_fixed(self, buf: np.ndarray) -> np.ndarray:
        """uint8[k] of every rank (same k) -> uint8[world, k]"""
        raise NotImplementedError

    def close(self):
        pass

    # ---- built on all_gather_fixed
    def all_gather_bytes(self, blob: bytes) -> list:
        """A byte string of any length from every rank, to every rank (one size exchange + one
        padded all-gather)."""
        mine = np.array([len(blob)], dtype=np.int64)
        sizes = self.all_gather_fixed(mine.view(np.uint8)).view(np.int64).reshape(-1)
        cap = max(int(sizes.max()), 1)
        cap = (cap + 15) // 16 * 16
        send = np.zeros(cap, dtype=np.uint8)
        send[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
        recv = self.all_gather_fixed(send)
        return [recv[r, :int(sizes[r])].tobytes() for r in range(self.world)]

    def all_gather_records(self, local: np.ndarray) -> list:
        """A 1-D structured / plain numpy array whose length differs per rank -> per-rank arrays."""
        local = np.ascontiguousarray(local)
        blobs = self.all_gather_bytes(local.view(np.uint8).reshape(-1).tobytes())
        return [np.frombuffer(b, dtype=local.dtype).copy() for b in blobs]

    def gather_bytes(self, blob: bytes, dst=0):
        """A byte string of any length from every rank, on rank `dst` only (None elsewhere).  The
        default goes through the all-gather; RcclComm sends to `dst` alone."""
        blobs = self.all_gather_bytes(blob)
        return blobs if self.rank == dst else None

    def gather_objects(self, obj, dst=0):
        """Small python metadata (names, comment lines) of every rank, on rank `dst`."""
        blobs = self.all_gather_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        return [pickle.loads(b) for b in blobs] if self.rank == dst else None


class SoloComm(Comm):
    """A world of one: lets one rank of a job run the single-process path in-process (bench.py measures rank 0 alone on the
    files the ranks have just shared out)."""

    def barrier(self):
        pass

    def max_float(self, x: float) -> float:
        return float(x)

    def all_gather_fixed(self, buf: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(buf, dtype=np.uint8).reshape(1, -1)


class RcclComm(Comm):
    """RCCL through the C ABI (pfa_comm_* in include/pfa.h).  The process is bound to GPU
    LOCAL_RANK first (`_lib.init`): one rule for the scan and the communicator."""

    def __init__(self):
        from . import _lib
        self._lib = _lib
        self.rank, self.local_rank, self.world = env_world()
        lib = _lib.load()
        _lib.init()
        n = _lib.PFA_COMM_ID_BYTES

        def make_id():
            buf = (ctypes.c_uint8 * n)()
            _lib.check(lib.pfa_comm_unique_id(buf))
            return bytes(buf)

        with stdout_to_stderr():
            uid = exchange_unique_id(self.rank, make_id, n)
            _lib.check(lib.pfa_comm_init(self.rank, self.world, (ctypes.c_uint8 * n).from_buffer_copy(uid)))
            _lib.check(lib.pfa_comm_barrier())     # builds the channels now (banner, lazy init), not mid-output
        if self.rank == 0 and not os.environ.get("PFA_RDZV_FILE"):
            with contextlib.suppress(OSError):     # every rank is past its read: the barrier above
                os.unlink(rendezvous_path())

    def barrier(self):
        self._lib.check(self._lib.load().pfa_comm_barrier())

    def max_float(self, x: float) -> float:
        v = ctypes.c_double(float(x))
        self._lib.check(self._lib.load().pfa_comm_all_reduce_max(ctypes.byref(v)))
        return float(v.value)

    def all_gather_fixed(self, buf: np.ndarray) -> np.ndarray:
        send = np.ascontiguousarray(buf, dtype=np.uint8).reshape(-1)
        recv = np.empty((self.world, len(send)), dtype=np.uint8)
        self._lib.check(self._lib.load().pfa_comm_all_gather(self._lib.ptr(send), self._lib.ptr(recv), len(send)))
        return recv

    def gather_bytes(self, blob: bytes, dst=0):
        """Variable-size gather to one rank: the sizes travel in a one-word all-gather, the payloads as
        ncclSend / ncclRecv to `dst` alone (pfa_comm_gatherv) — nobody but the printing rank receives
        the records, and nothing is padded to the largest rank."""
        lib = self._lib.load()
        mine = np.array([len(blob)], dtype=np.uint64)
        counts = np.ascontiguousarray(self.all_gather_fixed(mine.view(np.uint8)).view(np.uint64).reshape(-1))
        send = np.frombuffer(blob, dtype=np.uint8) if len(blob) else np.zeros(1, dtype=np.uint8)
        recv = np.empty(max(int(counts.sum()), 1), dtype=np.uint8) if self.rank == dst else None
        self._lib.check(lib.pfa_comm_gatherv(self._lib.ptr(np.ascontiguousarray(send)), len(blob), self._lib.ptr(recv),
                                             self._lib.ptr(counts), int(dst)))
        if self.rank != dst:
            return None
        ends = np.cumsum(counts).tolist()
        return [recv[e - int(c):e].tobytes() for e, c in zip(ends, counts.tolist())]

    def all_gather_plan_results(self, plan) -> np.ndarray:
        """The records of the plan's last run, of every rank, gathered from device memory
        (call before plan.fetch()): RESULT_DTYPE[world, n_units]."""
        out = np.empty((self.world, plan.table.n_units), dtype=self._lib.RESULT_DTYPE)
        self._lib.check(self._lib.load().pfa_comm_all_gather_results(plan.handle, self._lib.ptr(out)))
        return out

    def close(self):
        self._lib.load().pfa_comm_destroy()
