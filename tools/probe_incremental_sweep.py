"""Diagnostic: the incremental step over model shapes and hypothesis counts (split precision), to find shapes that fall off the
fast kernels.   python tools/probe_incremental_sweep.py [depth,width,voc,n_ctx ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib.engine import HipLM
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (2, 512, 256, 1), (1, 512, 256, 1), (3, 512, 256, 1), (4, 512, 256, 2), (2, 512, 100, 1), (2, 512, 1000, 1), (2, 256, 256, 1),
    (2, 128, 256, 1), (2, 64, 256, 1), (2, 1024, 256, 1), (2, 300, 256, 1), (2, 1100, 256, 1)]
ns = [int(x) for x in os.environ.get("KL_SWEEP_N", "1,17,100,128,250,256,600,1024").split(",")]
S = 120
for L, W, V, NC in shapes:
    lm = HipLM(L, W, V, NC)
    lm.init_weights(seed=4, emb_std=0.5)
    lm.prepare(3)
    rng = np.random.default_rng(3)
    line = []
    for n in ns:
        lm.ensure_pool(2 * n)
        ids = torch.from_numpy(rng.integers(1, V, size=(S, n)).astype(np.int32)).cuda()
        cc = torch.from_numpy(rng.integers(0, 200, size=(n, max(NC, 1))).astype(np.int32)).cuda()
        a = torch.arange(n, dtype=torch.int32).cuda(); b = a + n
        for s in range(20):
            lm.step_slots(ids[s], cc, a, b); a, b = b, a
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(S):
            lm.step_slots(ids[s], cc, a, b); a, b = b, a
        torch.cuda.synchronize()
        line.append("%d: %.1f" % (n, (time.perf_counter() - t0) / S * 1e6))
    print("depth %d width %4d V %4d contexts %d | us per step at n = %s" % (L, W, V, NC, "  ".join(line)), flush=True)
    del lm
    torch.cuda.empty_cache()
