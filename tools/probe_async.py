"""Diagnostic: does the host get control back while a training step runs?  (enqueue time, then the wait in read_loss
after 20 ms of simulated host work)"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 256
lm = HipLM(2, 512, 256, 1); lm.init_weights(seed=1); lm.prepare(hipabi.KL_PREC_BF16); lm.ensure_training_buffers(); lm.reset_states(B)
rng = np.random.default_rng(0)
x = rng.integers(1, 256, (B, T)).astype(np.int32); z = np.zeros((B, T, 1), np.int32); y = x.copy()
for k in range(8):
    t0 = time.perf_counter()
    masks = lm.draw_dropout_masks(B)
    t1 = time.perf_counter()
    lm.train_window(x, z, y, masks)
    t2 = time.perf_counter()
    lm.adam_step()
    t3 = time.perf_counter()
    time.sleep(0.020)
    t4 = time.perf_counter()
    lm.read_loss(reset=True)
    t5 = time.perf_counter()
    print(f"masks {1e3*(t1-t0):5.2f} ms | train_window enqueue {1e3*(t2-t1):5.2f} | adam enqueue {1e3*(t3-t2):5.2f} | read_loss wait after 20 ms of host work {1e3*(t5-t4):5.2f}")
