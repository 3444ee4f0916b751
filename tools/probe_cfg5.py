"""Diagnostic: training step and incremental step at the cfg5 topology (depth 4, width 1024, length 512, 2 contexts)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
L, W, V, T, C = 4, int(os.environ.get("KL_PROBE_W", "1024")), 256, 512, 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lm = HipLM(L, W, V, C)
lm.init_weights(seed=1)
lm.prepare(hipabi.KL_PREC_BF16)
rng = np.random.default_rng(0)
idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
ctx = torch.from_numpy(rng.integers(0, 200, (B, 1, C)).repeat(T, axis=1).astype(np.int32)).cuda()
tgt = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
masks = torch.from_numpy(lm.draw_dropout_masks(B)).cuda()
lm.reset_states(B)
def step():
    lm.train_window(idx, ctx, tgt, masks); lm.adam_step()
for _ in range(2): step()
torch.cuda.synchronize()
t = time.time(); n = 5
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.time() - t) / n
flops = 3.0 * B * T * sum(2.0 * ((W if l else W + 10 * C) + W) * 4 * W for l in range(L))
print(f"cfg5 train L={L} W={W} B={B} T={T}: {dt * 1e3:.1f} ms/step, {B * T / dt / 1e6:.3f} Mchars/s, {flops / dt / 1e12:.1f} TFLOP/s, loss {lm.read_loss()[0]:.3f}")
# incremental step (generate / rate_best) in split precision
n_h = 1024
lm.prepare(hipabi.KL_PREC_SPLIT)
lm.ensure_pool(2 * n_h)
a = torch.arange(n_h, dtype=torch.int32).cuda(); b = a + n_h
ii = torch.from_numpy(rng.integers(1, V, n_h).astype(np.int32)).cuda(); cc = torch.zeros((n_h, C), dtype=torch.int32).cuda()
for _ in range(5):
    lm.step_slots(ii, cc, a, b); a, b = b, a
torch.cuda.synchronize(); t = time.time(); k = 50
for _ in range(k):
    lm.step_slots(ii, cc, a, b); a, b = b, a
torch.cuda.synchronize(); dt = (time.time() - t) / k
print(f"cfg5 incremental step, {n_h} hypotheses, split precision: {dt * 1e6:.1f} us/step, {n_h / dt / 1e6:.2f} M hyp*chars/s")
