"""Stress the persistent scans: many consecutive windows, optional concurrent work on
torch's default stream; reports the first window whose hand-off timed out."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
concurrent = int(sys.argv[3]) if len(sys.argv) > 3 else 1
L, W, V, T = 2, 512, 256, 256
lm = HipLM(L, W, V, 1)
lm.init_weights(seed=1)
lm.prepare(hipabi.KL_PREC_BF16)
lm.ensure_training_buffers()
rng = np.random.default_rng(0)
idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
ctx = torch.zeros((B, T, 1), dtype=torch.int32).cuda()
gen = torch.Generator(device='cuda')
gen.manual_seed(2)
bad = 0
t0 = time.time()
for s in range(steps):
    masks = None
    if concurrent:
        keep = torch.rand((L, B, W), device='cuda', generator=gen) >= 0.1
        masks = keep.to(torch.float32) / 0.9
    lm.train_window(idx, ctx, idx, masks)
    lm.adam_step()
    if s % 5 == 4 or s == steps - 1:
        torch.cuda.synchronize()
        v = lm.loss_acc.cpu().numpy()
        if v[3] != 0:
            print(f"window <= {s}: hand-off timed out (loss_acc {v})", flush=True)
            bad += 1
            lm.loss_acc.zero_()
print(f"B={B} steps={steps} concurrent={concurrent}: {bad} bad checks, {time.time() - t0:.1f} s")
