"""Data-parallel training across the GPUs of one node (north_star; NOT a reference
feature -- the reference trains in one process on one device, rating.py:292-298).

One process per GPU, launched by `python -m torch.distributed.run`; each rank owns
`Rater.streams` independent stateful streams (its own files, carried states and
reset points); after the backward pass the flat f32 gradient vector (parameter
layout of include/keraslm_hip.h) is summed with ONE all-reduce over RCCL/xGMI
(`backend="nccl"` is RCCL on ROCm), then every rank applies the identical fused
clip+Adam update to 1/world of that sum (kl_adam_step_scaled: the mean costs no pass
of its own), so parameters stay bit-identical from step to step.

What makes them identical to begin with: rank 0's parameters (and Adam step count) are
broadcast after every re-definition of the model (`broadcast_params`), and rank 0's
shuffled file order / train-validation split is broadcast too (`broadcast_object`) --
`Rater.seed` defaults to None and the reference shuffles with the global `random`.
Whatever makes a rank leave the training loop (NaN loss, SIGINT, a failed hand-off) is
agreed on with a MAX all-reduce of flags (`any_flag`) before anybody leaves, so no rank
is left waiting in a collective.

The per-rank loss is already the mean over its B*T positions, so sum / world is the
mean over the global batch.  On CPU test doubles the same code runs over gloo.
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

    def reduce(self, lm):
        """all-reduce (SUM) of lm.grads in place; returns the scale that turns the sum into the mean over
        the ranks -- pass it to lm.adam_step(grad_scale=...)"""
        if self.world == 1:
            return 1.0
        g = lm.grads
        if hasattr(lm, "stream"):           # HIP engine: run the collective on the engine's stream
            import torch
            cur = torch.cuda.current_stream(lm.device)
            lm.stream.wait_stream(cur)
            with torch.cuda.stream(lm.stream):
                self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM)
            cur.wait_stream(lm.stream)
        else:
            self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM)
        return 1.0 / self.world

    def average(self, lm):
        """all-reduce (mean) of lm.grads in place (for callers that want the mean itself)"""
        scale = self.reduce(lm)
        if scale != 1.0:
            lm.grads.mul_(scale)

    def broadcast_params(self, lm):
        """rank 0's parameters and Adam step count to every rank (after configure / load_weights /
        reconfigure_for_mapping: each rank drew or loaded its own)"""
        if self.world == 1:
            return
        import torch
        if hasattr(lm, "params") and isinstance(lm.params, torch.Tensor):      # HIP engine: the flat device vector
            self.dist.broadcast(lm.params, src=0)
            t = torch.tensor([int(getattr(lm, "adam_t", 0))], dtype=torch.int64, device=lm.params.device)
            self.dist.broadcast(t, src=0)
            lm.adam_t = int(t.item())
            lm.prepare(lm.precision or 1)                                     # derived operands follow the parameters
        else:                                                                 # CPU test double: a dict of arrays
            w = lm.get_weights()
            names = sorted(w)
            flat = torch.from_numpy(__import__("numpy").concatenate([w[k].reshape(-1).astype("float32") for k in names]))
            self.dist.broadcast(flat, src=0)
            off = 0
            out = {}
            for k in names:
                n = w[k].size
                out[k] = flat[off:off + n].numpy().reshape(w[k].shape).astype(w[k].dtype)
                off += n
            lm.set_weights(out)

    def broadcast_object(self, obj):
        """rank 0's `obj` (picklable) on every rank"""
        if self.world == 1:
            return obj
        box = [obj if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def any_flag(self, *flags):
        """element-wise OR of boolean flags over the ranks (one small MAX all-reduce)"""
        if self.world == 1:
            return tuple(bool(f) for f in flags)
        import torch
        device = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return tuple(bool(x) for x in t.tolist())

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
