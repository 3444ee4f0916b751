"""Diagnostic: rating windows (Rater.rate / rate2 / test: the stateful windowed forward in split precision, probabilities left on
the device) over model shapes and stream counts.   python tools/probe_rating_sweep.py [depth,width,length,n_ctx ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (2, 512, 256, 1), (2, 512, 64, 1), (2, 512, 512, 1), (3, 512, 256, 1), (2, 256, 256, 1), (2, 128, 256, 1), (2, 64, 256, 1),
    (2, 1024, 256, 1), (4, 1024, 512, 2)]
Bs = [int(x) for x in os.environ.get("KL_SWEEP_B", "1,4,16,33,64,128,256,1024").split(",")]
V = 256
for L, W, T, C in shapes:
    lm = HipLM(L, W, V, C)
    lm.init_weights(seed=1)
    lm.prepare(hipabi.KL_PREC_SPLIT)
    rng = np.random.default_rng(0)
    line = []
    for B in Bs:
        try:
            idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
            ctx = torch.from_numpy(rng.integers(0, 200, (B, 1, max(C, 1))).repeat(T, axis=1).astype(np.int32)).cuda()
            lm.reset_states(B)
            for _ in range(3):
                lm.forward_window(idx, ctx)
            torch.cuda.synchronize()
            n = 10 if B * T < 100000 else 4
            t0 = time.perf_counter()
            for _ in range(n):
                lm.forward_window(idx, ctx)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            line.append("%d: %.2f ms (%.2f M/s)" % (B, dt * 1e3, B * T / dt / 1e6))
        except Exception as err:
            line.append("%d: %r" % (B, err))
    print("depth %d width %4d length %3d contexts %d | %s" % (L, W, T, C, "  ".join(line)), flush=True)
    del lm
    torch.cuda.empty_cache()
