"""The data-parallel HIP path on the one GPU of the test box: GradSync's engine-stream branch,
the device-tensor all-reduce between two processes and kl_adam_step_scaled, against ONE process that trains the
2B streams as a single batch.  (The CPU twin is tests/test_ddp_gloo.py; the real multi-GPU run is bench.py --gpus N.)"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import lstm_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(900)
@pytest.mark.parametrize("depth,width,voc,B,T", [(2, 128, 50, 16, 12),      # fused thin scans
                                                 (2, 512, 64, 256, 6)])     # layer-sequential wide scans, hipGraph replay
def test_two_ranks_on_one_gpu_equal_one_process(depth, width, voc, B, T):
    import torch
    from ocrd_keraslm_amd.lib import hipabi
    from ocrd_keraslm_amd.lib.engine import HipLM
    steps = 3
    cfg = O.ModelConfig(depth, width, voc, 1)
    w0 = O.init_weights(cfg, seed=4, emb_std=0.3)
    rng = np.random.default_rng(11)
    idx = rng.integers(0, voc, (steps, 2 * B, T)).astype(np.int32)
    ctx = rng.integers(0, 200, (steps, 2 * B, 1, 1)).repeat(T, axis=2).astype(np.int32)
    tgt = rng.integers(0, voc, (steps, 2 * B, T)).astype(np.int32)
    # ---- one process, 2B streams (the reference trajectory) + the second rank's gradients at that trajectory's weights
    ref = HipLM(depth, width, voc, 1)
    ref.set_weights(w0, hipabi.KL_PREC_BF16)
    ref.ensure_training_buffers()
    ref.reset_states(2 * B)
    half = HipLM(depth, width, voc, 1)
    half.set_weights(w0, hipabi.KL_PREC_BF16)
    half.ensure_training_buffers()
    half.reset_states(B)
    g1 = []
    for k in range(steps):
        half.set_weights(ref.get_weights(), hipabi.KL_PREC_BF16)      # (states of `half` carry on: set_weights leaves them)
        half.loss_acc.zero_()
        half.train_window(idx[k][B:], ctx[k][B:], tgt[k][B:], None)
        half.read_loss()
        g1.append(half.grads.detach().cpu().numpy().copy())
        ref.loss_acc.zero_()
        ref.train_window(idx[k], ctx[k], tgt[k], None)
        ref.adam_step()
        ref.read_loss()
    w_ref = ref.get_weights()
    st_ref = ref.get_states()
    del half
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as work:
        np.savez(os.path.join(work, "meta.npz"), depth=depth, width=width, voc=voc, B=B, T=T, steps=steps)
        np.savez(os.path.join(work, "weights0.npz"), **{"w_" + k: v for k, v in w0.items()})
        np.savez(os.path.join(work, "batches.npz"), idx=idx, ctx=ctx, tgt=tgt)
        np.save(os.path.join(work, "grads1.npy"), np.stack(g1))
        port = str(_free_port())
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_hip_worker.py"), str(r), "2", port, work],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                p.kill()
                out, _ = p.communicate()
            outs.append(out)
        for r, p in enumerate(procs):
            assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
        w_ddp = dict(np.load(os.path.join(work, "ddp_weights.npz")))
        st_ddp = np.load(os.path.join(work, "ddp_states.npy"))
    # The 2B-stream batch and the two B-stream halves sum in different orders (bf16 operands, split-K atomics).  Adam's
    # first steps move every weight by about +-lr whatever the size of its gradient, so an element whose gradient is
    # rounding noise may step the other way (2e-3 per step); everything else must agree closely.  A wrong scale, a
    # missed all-reduce or a stale gradient buffer would move ALL elements.
    for k in w_ref:
        d = np.abs(w_ddp[k] - w_ref[k])
        assert d.mean() < 1e-4, (k, d.mean())
        assert (d > 1e-3).mean() < 5e-3, (k, (d > 1e-3).mean())
        assert d.max() < 7e-3, (k, d.max())
    assert np.abs(st_ddp - st_ref[:B]).max() < 3e-2
