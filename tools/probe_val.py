"""Diagnostic: a validation window (forward only, bf16 operands, no probabilities back) against a training step."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 256
lm = HipLM(2, 512, 256, 1); lm.init_weights(seed=1); lm.prepare(hipabi.KL_PREC_BF16); lm.reset_states(B)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(1, 256, (B, T)).astype(np.int32)).cuda(); z = torch.zeros((B, T, 1), dtype=torch.int32).cuda()
for prec, name in ((hipabi.KL_PREC_BF16, "bf16"), (hipabi.KL_PREC_SPLIT, "split")):
    lm.prepare(prec)
    for _ in range(2): lm.forward_window(x, z, x, want_probs=False)
    torch.cuda.synchronize(); t = time.time(); n = 5
    for _ in range(n): lm.forward_window(x, z, x, want_probs=False)
    torch.cuda.synchronize(); dt = (time.time() - t) / n
    print(f"forward_window B={B} T={T} {name}: {dt * 1e3:.2f} ms, {B * T / dt / 1e6:.2f} Mchars/s")
