"""A dist.Comm over torch.distributed's gloo backend — TEST infrastructure only: it lets two
ranks exercise the product's multi-rank host path (sharding, gather, ordered merge) where
RCCL cannot run: on CPU, and with two ranks sharing the one GPU of the test box (RCCL
refuses two ranks on one device).  The product's communicator is dist.RcclComm."""
import numpy as np
import torch
import torch.distributed as td

from polyfasta_amd import dist


class GlooComm(dist.Comm):
    def __init__(self):
        self.rank, self.local_rank, self.world = dist.env_world()
        # torch's intra-op pool is one OpenMP thread per CPU, and they spin after every op: on a 256-CPU box with a CPU quota
        # of 16 the pools of four ranks froze the whole control group for most of every scheduler period (a 400 kB gather
        # took 0.3-1.8 s; 3-16 ms with one thread).  The product's communicator (RCCL) has no such pool.
        torch.set_num_threads(1)
        if not td.is_initialized():
            with dist.stdout_to_stderr():        # gloo announces its peers on stdout
                td.init_process_group(backend="gloo")
                td.barrier()

    def barrier(self):
        td.barrier()

    def max_float(self, x):
        t = torch.tensor([float(x)], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    def all_gather_fixed(self, buf):
        send = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.uint8).reshape(-1).copy())
        recv = torch.empty(self.world * send.numel(), dtype=torch.uint8)
        td.all_gather_into_tensor(recv, send)
        return recv.numpy().reshape(self.world, -1)

    def all_gather_plan_results(self, plan):
        """what RcclComm does on the device, through the host: the records of the plan's last run, of every rank"""
        import ctypes
        from polyfasta_amd import _lib
        ptr, nbytes = plan.results_dev()
        local = np.empty(nbytes, dtype=np.uint8)
        _lib.check(_lib.load().pfa_download(_lib.ptr(local), ctypes.c_void_p(ptr), nbytes))
        return self.all_gather_fixed(local).view(_lib.RESULT_DTYPE).reshape(self.world, -1)

    def close(self):
        if td.is_initialized():
            td.destroy_process_group()
