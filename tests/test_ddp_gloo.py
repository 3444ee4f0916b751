"""Data-parallel training path on CPU: world_size 2 over gloo (the GPU path uses the
same code over RCCL).  Each rank owns one stateful stream; gradients are averaged
with one all-reduce per step; the result must equal a single process training the
same two streams as one batch, and both ranks must end with identical weights."""
import os
import random
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ocrd_keraslm_amd.lib import Rater
from ocrd_keraslm_amd.lib.distributed import GradSync
from tests.oracle_engine import OracleLM

ALPHABET = "abcdefgh \n"


def write_corpus(tmp, n_files=4, size=150, seed=0):
    rng = np.random.default_rng(seed)
    names = []
    for i in range(n_files):
        # order-1 Markov-ish text
        idx = [int(rng.integers(len(ALPHABET)))]
        for _ in range(size - 1):
            idx.append((idx[-1] + int(rng.integers(1, 4))) % len(ALPHABET))
        name = os.path.join(tmp, "auth_title%d_%d.txt" % (i, 1700 + 10 * i))
        with open(name, "w") as f:
            f.write("".join(ALPHABET[j] for j in idx))
        names.append(name)
    return names


def train_once(names, streams, epochs=2, seed=0):
    if seed is not None:
        random.seed(seed)
    r = Rater(engine_factory=OracleLM)
    r.width, r.depth, r.length = 16, 1, 8
    r.stateful = True
    r.streams = streams
    r.max_epochs = epochs
    r.seed = seed
    r.char_degradation = 0.0
    r.context_degradation = 0.0
    r.configure()
    files = [open(n) for n in names[:-1]]
    val = [open(names[-1])]
    cwd = os.getcwd()
    os.chdir(os.path.dirname(names[0]))      # checkpoints land in the tmp dir
    try:
        r.train(files, val_data=val)
    finally:
        os.chdir(cwd)
    assert r.status == 2
    return r.model.get_weights(), r.history


def _worker(rank, world, port, names, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sync = GradSync()
        assert (sync.rank, sync.world) == (rank, world)
        # 1) the collective itself: mean over ranks
        class G:
            grads = torch.full((5,), float(rank + 1))
        sync.average(G)
        assert torch.allclose(G.grads, torch.full((5,), 1.5))
        a, b = sync.mean_scalars(float(rank), 2.0 * rank)
        assert abs(a - 0.5) < 1e-12 and abs(b - 1.0) < 1e-12
        G.grads = torch.full((5,), float(rank + 1))
        assert sync.reduce(G) == 0.5 and torch.allclose(G.grads, torch.full((5,), 3.0))      # the sum; the scale goes to Adam
        assert sync.any_flag(rank == 1, False, rank == 0) == (True, False, True)
        assert sync.broadcast_object(["order", rank]) == ["order", 0]
        # 2) the training loop, one stream per rank
        w, hist = train_once(names, streams=1)
        np.savez(os.path.join(out, "rank%d.npz" % rank), **w)
        np.save(os.path.join(out, "loss%d.npy" % rank), np.array(hist["loss"]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker_unseeded(rank, world, port, names, out):
    """What a user gets from `torchrun ... keraslm-rate train`: Rater.seed is None and the global `random` is
    seeded differently in every process -- initial weights, file order and dropout masks differ per rank unless
    the training loop makes them agree."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        random.seed(1000 + rank)
        np.random.seed(2000 + rank)
        w, hist = train_once(names, streams=1, seed=None)
        np.savez(os.path.join(out, "urank%d.npz" % rank), **w)
        np.save(os.path.join(out, "uval%d.npy" % rank), np.array(hist["val_loss"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_unseeded_ranks_end_with_identical_weights():
    """without a shared seed the ranks must still train ONE model -- rank 0's initial weights and
    file order are broadcast, so weights and validation losses are identical on both ranks afterwards."""
    with tempfile.TemporaryDirectory() as tmp:
        names = write_corpus(tmp)
        mp.spawn(_worker_unseeded, args=(2, _free_port(), names, tmp), nprocs=2, join=True)
        w0 = dict(np.load(os.path.join(tmp, "urank0.npz")))
        w1 = dict(np.load(os.path.join(tmp, "urank1.npz")))
        for k in w0:
            assert np.array_equal(w0[k], w1[k]), "ranks diverged on %s" % k
        assert np.array_equal(np.load(os.path.join(tmp, "uval0.npy")), np.load(os.path.join(tmp, "uval1.npy")))


@pytest.mark.timeout(600)
def test_two_ranks_equal_one_process_with_two_streams():
    with tempfile.TemporaryDirectory() as tmp:
        names = write_corpus(tmp)
        mp.spawn(_worker, args=(2, _free_port(), names, tmp), nprocs=2, join=True)
        w0 = dict(np.load(os.path.join(tmp, "rank0.npz")))
        w1 = dict(np.load(os.path.join(tmp, "rank1.npz")))
        for k in w0:
            assert np.array_equal(w0[k], w1[k]), "ranks diverged on %s" % k
        ref, hist = train_once(names, streams=2)
        for k in w0:
            assert np.abs(w0[k] - ref[k]).max() < 1e-5, k
        # each rank logs the loss of ITS stream; their mean is the two-stream batch loss
        l0 = np.load(os.path.join(tmp, "loss0.npy"))
        l1 = np.load(os.path.join(tmp, "loss1.npy"))
        assert np.abs((l0 + l1) / 2 - np.array(hist["loss"])).max() < 1e-4
