"""Re-run bench.py's training loop at B streams with a status check every step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sync_every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = 256
lm = HipLM(2, 512, 256, 1); lm.init_weights(seed=1); lm.prepare(1); lm.ensure_training_buffers()
corpus = bench.synthetic_corpus()
per = bench.CORPUS // B
streams = torch.from_numpy(np.stack([corpus[s * per:(s + 1) * per] for s in range(B)])).cuda()
rng = np.random.default_rng(7)
ctx = torch.from_numpy(rng.integers(0, 200, size=B).astype(np.int32)).cuda()[:, None, None].expand(B, T, 1).contiguous()
gen = torch.Generator(device='cuda'); gen.manual_seed(2)
lm.reset_states(B)
for w in range(40):
    idx = streams[:, w * T:(w + 1) * T].contiguous(); tgt = streams[:, w * T + 1:(w + 1) * T + 1].contiguous()
    masks = (torch.rand((2, B, 512), device='cuda', generator=gen) >= 0.1).to(torch.float32) / 0.9
    lm.train_window(idx, ctx, tgt, masks); lm.adam_step()
    if w % sync_every == sync_every - 1:
        torch.cuda.synchronize()
        v = lm.loss_acc.cpu().numpy().copy(); lm.loss_acc.zero_()
        print(w, v, flush=True)
