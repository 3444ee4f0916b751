"""One rank of tests/test_ddp_hip.py (started as a fresh child process, never by re-executing pytest).

Rank 0 owns the GPU work: a HipLM with B streams trains `steps` windows; after every backward pass its flat gradient
vector goes through distributed.GradSync.reduce -- the engine-stream branch, an all-reduce of a device tensor over
gloo -- and kl_adam_step_scaled applies 1/world of the sum.  Rank 1 never launches a persistent scan (two ranks
scanning on ONE GPU would break the scans' co-residency): it contributes the gradients of "its" B streams, which the
test computed beforehand in a single process, through the same collective."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, work = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    meta = dict(np.load(os.path.join(work, "meta.npz")))
    depth, width, voc, B, T, steps = (int(meta[k]) for k in ("depth", "width", "voc", "B", "T", "steps"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ocrd_keraslm_amd.lib.distributed import GradSync
        sync = GradSync()
        assert (sync.rank, sync.world) == (rank, world)
        dev = torch.device("cuda:0")
        # the collective accepts device tensors on both sides (checked on 5 elements before anything depends on it)
        probe = torch.full((5,), float(rank + 1), device=dev)
        dist.all_reduce(probe)
        assert torch.allclose(probe.cpu(), torch.full((5,), 3.0)), probe
        if rank == 0:
            from ocrd_keraslm_amd.lib import hipabi
            from ocrd_keraslm_amd.lib.engine import HipLM
            lm = HipLM(depth, width, voc, 1)
            w = {k[2:]: v for k, v in np.load(os.path.join(work, "weights0.npz")).items()}
            lm.set_weights(w, hipabi.KL_PREC_BF16)
            lm.ensure_training_buffers()
            lm.reset_states(B)
            sync.broadcast_params(lm)                      # (rank 1 joins with a same-sized dummy: see below)
            data = np.load(os.path.join(work, "batches.npz"))
            for k in range(steps):
                lm.loss_acc.zero_()
                lm.train_window(data["idx"][k][:B], data["ctx"][k][:B], data["tgt"][k][:B], None)
                scale = sync.reduce(lm)
                assert scale == 0.5
                lm.adam_step(grad_scale=scale)
                lm.read_loss()
                flags = sync.any_flag(False, False, False)
                assert flags == (False, False, False)
            np.savez(os.path.join(work, "ddp_weights.npz"), **lm.get_weights())
            np.save(os.path.join(work, "ddp_states.npy"), lm.get_states())
        else:
            g1 = np.load(os.path.join(work, "grads1.npy"))          # [steps][n_params]
            class Dummy:      # what broadcast_params needs from an engine: a flat device vector and a step count
                params = torch.zeros(g1.shape[1], dtype=torch.float32, device=dev)
                adam_t = 0
                precision = 1
                def prepare(self, _p):
                    pass
            sync.broadcast_params(Dummy())
            for k in range(steps):
                class G:
                    grads = torch.from_numpy(g1[k]).to(dev)
                sync.reduce(G)
                sync.any_flag(False, False, False)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
