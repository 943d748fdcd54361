"""One process per GPU: sharding of independent alignments and the final gather.

The `--dir` batch mode of the reference is an embarrassingly parallel loop over
files (PolyFastA.py:104).  Here the sorted file list is dealt round-robin to
the ranks of a `torch.distributed` job (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests); no collective touches the data path.  When every
rank has scanned its share, ONE all-gather moves the fixed-size integer result
records (and the small coding-sequence bins) to every rank; rank 0 finishes the
floats and prints the rows in the reference's order.

torch is used for the process group only (rendezvous + collectives): the scan
itself never sees a torch type.
"""
from __future__ import annotations

import os

import numpy as np


import contextlib
import sys


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on stdout when the first communicator is built;
    stdout carries the CSV rows (or the bench's one JSON line), so file descriptor 1
    is pointed at stderr while the process group comes up."""
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
    """(rank, local_rank, world) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(n_items: int, rank: int, world: int):
    """Indices of the items rank `rank` owns: round-robin, so that neighbouring
    (similarly sized) files spread over all GPUs."""
    return list(range(rank, n_items, world))


class Group:
    """Thin wrapper over a torch.distributed process group."""

    def __init__(self, backend=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.local_rank, self.world = env_world()
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        self.backend = backend
        self.device = torch.device("cuda", self.local_rank) if backend == "nccl" else torch.device("cpu")
        if not dist.is_initialized():
            with stdout_to_stderr():
                if backend == "nccl":
                    torch.cuda.set_device(self.local_rank)
                    dist.init_process_group(backend="nccl", device_id=self.device)
                else:
                    dist.init_process_group(backend="gloo")
                self.barrier()          # builds the communicator now (banner, lazy init) rather than mid-output

    def barrier(self):
        if self.backend == "nccl":
            self.torch.cuda.synchronize()
        self.dist.barrier()

    def all_gather_records(self, local: np.ndarray) -> list:
        """All-gather a 1-D structured/plain numpy array whose length differs per
        rank; returns the list of per-rank arrays (same dtype).  One size
        exchange + one padded all_gather_into_tensor."""
        torch, dist = self.torch, self.dist
        local = np.ascontiguousarray(local)
        item = local.dtype.itemsize
        counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(counts, mine)
        counts = counts.cpu().tolist()
        cap = max(max(counts), 1) * item
        buf = np.zeros(cap, dtype=np.uint8)
        buf[:local.nbytes] = local.view(np.uint8).reshape(-1)
        send = torch.from_numpy(buf).to(self.device)
        recv = torch.empty(self.world * cap, dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(recv, send)
        if self.backend == "nccl":
            torch.cuda.synchronize()
        flat = recv.cpu().numpy()
        return [flat[r * cap:r * cap + counts[r] * item].view(local.dtype).copy() for r in range(self.world)]

    def gather_objects(self, obj, dst=0):
        """Small python metadata (names, comment lines) to rank `dst`."""
        out = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(obj, out, dst=dst)
        return out

    def max_float(self, x: float) -> float:
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
