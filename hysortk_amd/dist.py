"""Process-group plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on
ROCm, "gloo" on CPU) for the small host-side collectives the reference does with MPI on its
MPI_Comm (Exscan of read counts, histogram Allreduce, barriers).  The supermer payload itself
moves inside libhsk.so with RCCL send/recv (csrc/hsk_comm.h), not through this module."""
import os

import numpy as np


class Comm:
    def __init__(self, backend=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        self.backend = backend
        if backend == "nccl":
            torch.cuda.set_device(self.local_rank if device is None else device)
            self.dev = torch.device("cuda", self.local_rank if device is None else device)
        else:
            self.dev = torch.device("cpu")
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.size)

    def barrier(self):
        if self.size > 1:
            self.dist.barrier()

    def allreduce_sum(self, arr):
        a = np.ascontiguousarray(arr)
        t = self.torch.from_numpy(a.astype(np.int64)).to(self.dev)
        if self.size > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().astype(a.dtype)

    def allreduce_max(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.dev)
        if self.size > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather_i64(self, value):
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device=self.dev)
        out = [self.torch.zeros_like(t) for _ in range(self.size)]
        if self.size > 1:
            self.dist.all_gather(out, t)
        else:
            out = [t]
        return [int(x.item()) for x in out]

    def exscan_sum(self, value):
        """MPI_Exscan(SUM); rank 0 gets 0 (reference src/kmerops.cpp:65-71)."""
        return sum(self.allgather_i64(value)[: self.rank])

    def bcast_bytes(self, raw, root=0):
        t = self.torch.tensor(list(raw), dtype=self.torch.uint8, device=self.dev)
        if self.size > 1:
            self.dist.broadcast(t, src=root)
        return bytes(t.cpu().numpy().tobytes())

    def destroy(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
