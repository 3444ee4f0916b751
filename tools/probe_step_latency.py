"""Host-side latency of ONE incremental step as Rater.predict issues it (rating.py:578-639): the call itself, and the call
plus the copy of the probabilities to the host -- the GPU idle in between, as in a beam search.
  python tools/probe_step_latency.py [n ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib.engine import HipLM
L, W, V = 2, 512, 256
lm = HipLM(L, W, V, 1)
lm.init_weights(seed=4, emb_std=0.5)
lm.prepare(3)
rng = np.random.default_rng(3)
for n in [int(a) for a in sys.argv[1:]] or [30, 128]:
    lm.ensure_pool(2 * n)
    a = np.arange(n, dtype=np.int32); b = a + n
    cc = rng.integers(0, 200, size=(n, 1)).astype(np.int32)
    t_call, t_all, t_h2d = [], [], []
    for s in range(300):
        ids = rng.integers(1, V, size=n).astype(np.int32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids_d, cc_d, a_d, b_d = lm.to_device_i32(ids), lm.to_device_i32(cc), lm.to_device_i32(a), lm.to_device_i32(b)
        t1 = time.perf_counter()
        p = lm.step_slots(ids_d, cc_d, a_d, b_d)
        t2 = time.perf_counter()
        ph = p.cpu()
        t3 = time.perf_counter()
        a, b = b, a
        if s >= 50:
            t_h2d.append(t1 - t0); t_call.append(t2 - t1); t_all.append(t3 - t0)
    f = lambda v: f"{np.median(v) * 1e6:7.1f} us (p90 {np.percentile(v, 90) * 1e6:7.1f})"
    print(f"n={n:5d}: four index copies to the device {f(t_h2d)}, step_slots call {f(t_call)}, whole step with probabilities on the host {f(t_all)}")
