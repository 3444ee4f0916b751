"""Where do the carried states of the second-generation scans differ from the first generation's?"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
depth, width, voc, B, T = [int(x) for x in sys.argv[1:6]]
n_ctx = int(sys.argv[6]) if len(sys.argv) > 6 else 1
from oracle import lstm_oracle as O
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
cfg = O.ModelConfig(depth, width, voc, n_ctx)
w = O.init_weights(cfg, seed=4, emb_std=0.3)
rng = np.random.default_rng(21)
idx = rng.integers(0, voc, (B, T)); ctx = rng.integers(0, 200, (B, 1, max(n_ctx, 1))).repeat(T, axis=1)[:, :, :n_ctx]; tgt = rng.integers(0, voc, (B, T))
st0 = [rng.standard_normal((B, width)) * 0.1 for _ in range(2 * depth)]
states = np.stack(st0, axis=1).astype(np.float32)
out = []
for flag in ("0", "1"):
    os.environ["KL_SCAN2"] = flag
    lm = HipLM(depth, width, voc, n_ctx)
    lm.set_weights(w, hipabi.KL_PREC_BF16)
    lm.reset_states(B); lm.set_states(states)
    lm.loss_acc.zero_()
    lm.train_window(idx, ctx, tgt, None)
    print("KL_SCAN2=" + flag, lm.read_loss())
    out.append((lm.get_states().copy(), {k: v.copy() for k, v in lm.get_grads().items()}))
a, b = out[0][0], out[1][0]
for k in range(2 * depth):
    d = np.abs(a[:, k] - b[:, k])
    rows = np.where(d.max(axis=1) > 5e-3)[0]
    units = np.where(d.max(axis=0) > 5e-3)[0]
    print(f"state {k} ({'hc'[k & 1]}{k // 2}): max diff {d.max():.4f}; rows off: {len(rows)} {rows[:40]}; units off: {len(units)} {units[:40]}")
for name in out[0][1]:
    ga, gb = out[0][1][name], out[1][1][name]
    d = np.abs(ga - gb) / (np.abs(ga).max() + 1e-12)
    print(f"grad {name}: max rel diff {d.max():.4f} at {np.unravel_index(d.argmax(), d.shape)}")
