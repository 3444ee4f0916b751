"""Incremental step (Rater.predict's kernel path, rating.py:578-639): wall and GPU-only time per chained step.
  python tools/probe_incremental.py [n ...]      (KL_PROBE_PREC=1|3, KL_PROBE_STEPS)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
L, W, V = int(os.environ.get('KL_PROBE_L', '2')), int(os.environ.get('KL_PROBE_W', '512')), 256
NC = int(os.environ.get('KL_PROBE_C', '1'))
S = int(os.environ.get("KL_PROBE_STEPS", "200"))
lm = HipLM(L, W, V, NC)
lm.init_weights(seed=4, emb_std=0.5)
rng = np.random.default_rng(3)
for prec in [int(x) for x in os.environ.get("KL_PROBE_PREC", "3,1").split(",")]:
    lm.prepare(prec)
    for n in [int(a) for a in sys.argv[1:]] or [128, 1024]:
        lm.ensure_pool(2 * n)
        ids = torch.from_numpy(rng.integers(1, V, size=(S, n)).astype(np.int32)).cuda()
        cc = torch.from_numpy(rng.integers(0, 200, size=(n, NC)).astype(np.int32)).cuda()
        a = torch.arange(n, dtype=torch.int32).cuda(); b = a + n
        for s in range(20):
            lm.step_slots(ids[s], cc, a, b); a, b = b, a
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for s in range(S):
            lm.step_slots(ids[s], cc, a, b); a, b = b, a
        e1.record(); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / S * 1e6
        gpu = e0.elapsed_time(e1) / S * 1e3
        print(f"n={n:5d} prec={prec}: wall {wall:7.1f} us/step, between events {gpu:7.1f} us/step, {n / wall:7.2f} M hyp*chars/s")
        if hasattr(lm, "step_chain"):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); e0.record()
            out = lm.step_chain(ids, cc, a, b)
            e1.record(); torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / S * 1e6
            print(f"         chained call: wall {wall:7.1f} us/step, between events {e0.elapsed_time(e1) / S * 1e3:7.1f} us/step")
