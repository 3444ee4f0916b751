"""Data-parallel training across the GPUs of one node (north_star; NOT a reference
feature -- the reference trains in one process on one device, rating.py:292-298).

One process per GPU, launched by `python -m torch.distributed.run`; each rank owns
`Rater.streams` independent stateful streams (its own files, carried states and
reset points); after the backward pass the flat f32 gradient vector (parameter
layout of include/keraslm_hip.h) is averaged with ONE all-reduce over RCCL/xGMI
(`backend="nccl"` is RCCL on ROCm), then every rank applies the identical fused
clip+Adam update, so parameters stay bit-identical without broadcasts.

The sum is taken over ranks and scaled by 1/world: the per-rank loss is already
the mean over its B*T positions, so the result is the mean over the global batch.
On CPU test doubles the same code runs over gloo.
"""
from __future__ import annotations

import os


class GradSync(object):
    def __init__(self):
        self.dist = None
        self.rank = 0
        self.world = 1
        try:
            import torch.distributed as dist
        except ImportError:      # pragma: no cover
            return
        if dist.is_available() and dist.is_initialized():
            self.dist = dist
            self.rank = dist.get_rank()
            self.world = dist.get_world_size()

    def average(self, lm):
        """all-reduce (mean) of lm.grads in place"""
        if self.world == 1:
            return
        g = lm.grads
        if hasattr(lm, "stream"):           # HIP engine: run the collective on the engine's stream
            import torch
            cur = torch.cuda.current_stream(lm.device)
            lm.stream.wait_stream(cur)
            with torch.cuda.stream(lm.stream):
                self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM)
                g.mul_(1.0 / self.world)
            cur.wait_stream(lm.stream)
        else:
            self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM)
            g.mul_(1.0 / self.world)

    def mean_scalars(self, *values):
        if self.world == 1:
            return values
        import torch
        device = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(values, dtype=torch.float64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return tuple((t / self.world).tolist())


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
