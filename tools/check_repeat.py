"""Repeatability of a training window at full size: the same window (inputs, carried state, dropout masks, weights) is run
N times and every run's loss, carried state and flat gradient vector are compared with the first run's.  The persistent
scans hand tiles between workgroups through counted waits, sentinels and flags; a tile taken too early (stale data) would
show up here as a run that differs by far more than the f32-atomics noise of the split-K weight-gradient GEMMs.
Usage: python tools/check_repeat.py [cfg2|cfg5|w128] [runs]      -> one JSON line"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM

what = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# (w128: the published model's size with both layers' scans in one launch, a layer polling its neighbour's rows)
L, W, V, T, C, B = {"cfg2": (2, 512, 256, 256, 1, 3072), "cfg5": (4, 1024, 256, 512, 2, 512), "w128": (2, 128, 256, 256, 1, 1024)}[what]
lm = HipLM(L, W, V, C)
lm.init_weights(seed=1, emb_std=0.3)
lm.prepare(hipabi.KL_PREC_BF16)
lm.ensure_training_buffers()
rng = np.random.default_rng(3)
idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
tgt = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
ctx = torch.from_numpy(rng.integers(0, 200, (B, 1, C)).repeat(T, axis=1).astype(np.int32)).cuda()
st0 = (rng.standard_normal((B, 2 * L, W)) * 0.3).astype(np.float32)
masks = lm.draw_dropout_masks(B)
ref = None
worst = {"grad": 0.0, "state": 0.0, "loss": 0.0}
status_bad = 0
for r in range(runs):
    lm.set_states(st0)
    lm.loss_acc.zero_()
    lm.train_window(idx, ctx, tgt, masks)
    torch.cuda.synchronize()
    acc = lm.loss_acc.cpu().numpy().copy()
    if acc[3] != 0:
        status_bad += 1
    g = lm.grads.detach().float().cpu().numpy().copy() if hasattr(lm, "grads") else np.concatenate([v.ravel() for v in lm.get_grads().values()])
    s = lm.get_states()
    loss = float(acc[0])
    if ref is None:
        ref = (g, s, loss)
        continue
    worst["grad"] = max(worst["grad"], float(np.abs(g - ref[0]).max() / max(1e-30, np.abs(ref[0]).max())))
    worst["state"] = max(worst["state"], float(np.abs(s - ref[1]).max()))
    worst["loss"] = max(worst["loss"], abs(loss - ref[2]) / max(1e-30, abs(ref[2])))
print(json.dumps({"workload": what, "streams": B, "seq_len": T, "runs": runs, "hand_off_time_outs": status_bad,
                  "grad_max_abs_diff_over_max": worst["grad"], "state_max_abs_diff": worst["state"],
                  "loss_rel_diff": worst["loss"],
                  "note": "run 0 against runs 1..N-1 of the SAME window; differences come from the f32 atomics of the split-K weight-gradient GEMMs only"}))
