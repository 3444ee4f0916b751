import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
L, W, V = 2, 512, 256
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lm = HipLM(L, W, V, 1); lm.init_weights(seed=1); lm.prepare(3)
lm.ensure_pool(2 * n)
a = torch.arange(n, dtype=torch.int32).cuda(); b = a + n
ii = torch.randint(1, V, (n,), dtype=torch.int32).cuda(); cc = torch.zeros((n, 1), dtype=torch.int32).cuda()
for _ in range(10): lm.step_slots(ii, cc, a, b); a, b = b, a
torch.cuda.synchronize(); t = time.time(); k = 200
for _ in range(k): lm.step_slots(ii, cc, a, b); a, b = b, a
torch.cuda.synchronize(); print(f"n={n}: {(time.time()-t)/k*1e6:.1f} us/step wall")
