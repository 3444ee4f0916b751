"""Per-array gradient errors (vs the f64 oracle) of one training window, first- vs second-generation wide scans."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
depth, width, voc, B, T = [int(x) for x in sys.argv[1:6]]
n_ctx = int(sys.argv[6]) if len(sys.argv) > 6 else 1
use_masks = (int(sys.argv[7]) if len(sys.argv) > 7 else 1) != 0
from oracle import lstm_oracle as O
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
cfg = O.ModelConfig(depth, width, voc, n_ctx)
w = O.init_weights(cfg, seed=4, emb_std=0.3)
rng = np.random.default_rng(21)
w64 = {k: v.astype(np.float64) for k, v in w.items()}
idx = rng.integers(0, voc, (B, T)); ctx = rng.integers(0, 200, (B, 1, n_ctx)).repeat(T, axis=1); tgt = rng.integers(0, voc, (B, T))
st0 = [rng.standard_normal((B, width)) * 0.1 for _ in range(2 * depth)]
states = np.stack(st0, axis=1).astype(np.float32)
res = {}
for flag in ("0", "1"):
    os.environ["KL_SCAN2"] = flag
    lm = HipLM(depth, width, voc, n_ctx)
    lm.set_weights(w, hipabi.KL_PREC_BF16)
    lm.reset_states(B); lm.set_states(states)
    if flag == "0":
        masks = lm.draw_dropout_masks(B) if use_masks else None
        om = [None] + [masks[l].astype(np.float64) for l in range(1, depth)] if use_masks else None
        ref_p, ref_st, cache = O.forward_window(cfg, w64, idx, ctx, [s.astype(np.float64) for s in st0], om, keep_cache=True)
        g_ref = O.backward_window(cfg, w64, idx, ctx, tgt, ref_p, cache, om)
    lm.loss_acc.zero_()
    lm.train_window(idx, ctx, tgt, masks)
    print("KL_SCAN2=" + flag, lm.read_loss())
    grads = lm.get_grads()
    for name, _o, _r, _c in lm.layout:
        got = grads[name].reshape(g_ref[name].shape)
        res.setdefault(name, []).append(np.abs(got - g_ref[name]).max() / (np.abs(g_ref[name]).max() + 1e-12))
        res.setdefault(name + ' relL2', []).append(np.linalg.norm(got - g_ref[name]) / (np.linalg.norm(g_ref[name]) + 1e-30))
    st = lm.get_states()
    res.setdefault("states", []).append(max(np.abs(st[:, k] - ref_st[k]).max() for k in range(2 * depth)))
for k, v in res.items():
    print(f"  {k:8s} gen1 {v[0]:.4f}  gen2 {v[1]:.4f}")
