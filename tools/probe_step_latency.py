"""Host-side latency of ONE incremental step as a beam search issues it (rating.py:578-639, 809-826): the GPU idle in
between, wall time from the call until the numbers are in the caller's hands.
  device-pointer entry (kl_step_batch): index copies + step_slots + copy of the probabilities to the host
  host entry (kl_step_batch_host):      HipLM.step_host -- whole rows / the per-row target probability / with head vectors
  python tools/probe_step_latency.py [--json] [n ...]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib.engine import HipLM


def run(ns=(30, 128), L=2, W=512, V=256, steps=300, skip=50):
    lm = HipLM(L, W, V, 1)
    lm.init_weights(seed=4, emb_std=0.5)
    lm.prepare(3)
    rng = np.random.default_rng(3)
    out = {}
    for n in ns:
        lm.ensure_pool(2 * n)
        cc = rng.integers(0, 200, size=(n, 1)).astype(np.int32)
        res = {}
        # device-pointer entry point, as Rater._predict_refs used it in round 3: one packed transfer, step, copy back
        a = np.arange(n, dtype=np.int32); b = a + n
        t_all = []
        for s in range(steps):
            ids = rng.integers(1, V, size=n).astype(np.int32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev = lm.to_device_i32(np.stack([ids, a, b, cc[:, 0]]))
            p = lm.step_slots(dev[0], dev[3], dev[1], dev[2]).cpu()
            t_all.append(time.perf_counter() - t0)
            a, b = b, a
        res["device_entry_us"] = float(np.median(t_all[skip:]) * 1e6)
        for name, kw in (("host_entry_rows_us", {}), ("host_entry_target_us", {"target": True}),
                         ("host_entry_target_heads_us", {"target": True, "head_k": L})):
            t_all = []
            for s in range(steps):
                ids = rng.integers(1, V, size=n).astype(np.int32)
                tg = rng.integers(0, V, size=n).astype(np.int32) if kw.get("target") else None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lm.step_host(ids, cc, a, b, target=tg, head_k=kw.get("head_k", 0))
                t_all.append(time.perf_counter() - t0)
                a, b = b, a
            res[name] = float(np.median(t_all[skip:]) * 1e6)
            res[name.replace("_us", "_p90_us")] = float(np.percentile(t_all[skip:], 90) * 1e6)
        out["n%d" % n] = res
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--json"]
    r = run([int(a) for a in args] or [30, 128])
    if "--json" in sys.argv:
        print(json.dumps(r))
    else:
        for k, v in r.items():
            print(k, {kk: round(vv, 1) for kk, vv in v.items()})
