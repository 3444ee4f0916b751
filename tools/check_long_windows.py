"""Diagnostic: the longest windows the reference allows (length 1024) at large batches -- shapes whose buffers pass
4 GiB, where the persistent scans (unsigned 32-bit buffer offsets) must step aside.  Loss against a second run through
the launch-per-step kernels (KL_SCAN=0) of the same build."""
import os, subprocess, sys
if len(sys.argv) > 1:
    import numpy as np, torch
    sys.path.insert(0, '.')
    from ocrd_keraslm_amd.lib import hipabi
    from ocrd_keraslm_amd.lib.engine import HipLM
    L, W, V, B, T = [int(x) for x in sys.argv[1:6]]
    lm = HipLM(L, W, V, 1); lm.init_weights(seed=1, emb_std=0.3); lm.prepare(hipabi.KL_PREC_BF16)
    rng = np.random.default_rng(0)
    idx = rng.integers(1, V, (B, T)).astype(np.int32); ctx = np.zeros((B, T, 1), np.int32)
    lm.reset_states(B)
    lm.train_window(idx, ctx, idx, None)
    l = lm.read_loss()
    g = lm.get_grads()
    print("RESULT %.6f %.6e %.6e" % (l[0], float(np.abs(g["U0"]).sum()), float(np.abs(g["E"]).sum())))
    sys.exit(0)
for shape in [(2, 512, 64, 1024, 1024), (2, 1024, 64, 1024, 512), (2, 1024, 64, 520, 1024)]:
    out = []
    for scan in ("1", "0"):
        env = dict(os.environ, KL_SCAN=scan)
        r = subprocess.run([sys.executable, __file__] + [str(x) for x in shape], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        out.append(line[0] if line else "FAILED: " + r.stderr[-300:])
    print(shape, "| scans:", out[0], "| per-step:", out[1], flush=True)
