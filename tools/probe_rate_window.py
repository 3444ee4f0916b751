"""Diagnostic: the rating window (Rater.rate: stateful windowed forward, split precision, probabilities out) for the
rocprofv3 kernel stats committed under profiles/ (tools/profile_round.sh)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM

if len(sys.argv) > 1 and sys.argv[1] == "cfg5":      # the cfg5 topology at several stream counts
    for B in [int(x) for x in sys.argv[2:]] or [1]:
        L, W, V, T, C = 4, 1024, 256, 512, 2
        lm = HipLM(L, W, V, C)
        lm.init_weights(seed=1)
        lm.prepare(hipabi.KL_PREC_SPLIT)
        rng = np.random.default_rng(0)
        idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
        ctx = torch.from_numpy(rng.integers(0, 200, (B, 1, C)).repeat(T, axis=1).astype(np.int32)).cuda()
        lm.reset_states(B)
        for _ in range(3):
            lm.forward_window(idx, ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            lm.forward_window(idx, ctx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("cfg5 rating window B=%d T=%d split precision: %.2f ms/window, %.1f k chars/s" % (B, T, dt * 1e3, B * T / dt / 1e3))
        del lm
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L, W, V, T = 2, 512, 256, 256
lm = HipLM(L, W, V, 1)
lm.init_weights(seed=1)
lm.prepare(hipabi.KL_PREC_SPLIT)
rng = np.random.default_rng(0)
idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
ctx = torch.from_numpy(rng.integers(0, 200, (B, 1, 1)).repeat(T, axis=1).astype(np.int32)).cuda()
lm.reset_states(B)
for _ in range(5):
    lm.forward_window(idx, ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 100
for _ in range(n):
    lm.forward_window(idx, ctx)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("rating window B=%d T=%d split precision: %.3f ms/window, %.1f k chars/s" % (B, T, dt * 1e3, B * T / dt / 1e3))
