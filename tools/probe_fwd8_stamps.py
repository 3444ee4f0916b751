"""Diagnostic: cycle shares of the eight-wave forward scan's phases (lstm_scan_fwd8.hip), wave 0 of workgroup 0.
Needs the stamps build: make -C ocrd_keraslm_amd/csrc stamps
  python tools/probe_fwd8_stamps.py [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
if not os.environ.get('KL_LIB'): hipabi.LIB_PATH = os.path.join(os.path.dirname(hipabi.LIB_PATH), os.environ.get('KL_STAMPS_LIB', 'libkeraslm_hip_stamps.so'))
from ocrd_keraslm_amd.lib.engine import HipLM

lib = hipabi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
L, W, V, T = 2, 512, 256, 256
lm = HipLM(L, W, V, 1)
lm.init_weights(seed=1)
lm.prepare(1)
rng = np.random.default_rng(0)
idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
ctx = torch.zeros((B, T, 1), dtype=torch.int32).cuda()
masks = lm.draw_dropout_masks_device(B)
for _ in range(3):
    lm.train_window(idx, ctx, idx, masks)
torch.cuda.synchronize()
lib.kl_test_fwd8_stamps.restype = C.c_int
lib.kl_test_fwd8_stamps(None, 1)
n = 4
for _ in range(n):
    lm.train_window(idx, ctx, idx, masks)
torch.cuda.synchronize()
st = (C.c_ulonglong * 32)()
lib.kl_test_fwd8_stamps(st, 0)
st = list(st)
np_f = B // 32 // 32
per = n * T * L * np_f
names = ['loop top (+ tile request at the top)', 'own rows landed + armed counter', 'MFMA phase + release', 'tile request + gate-input wait + next pieces',
         'epilogue math, G stores, staging', '(publisher) strips + arrival', '(publisher) publish + re-arm', 'strips + arrival / rotate']
v = np.array(st[:8], dtype=np.float64) / per
v[1] += st[8] / per; v[3] += (st[9] + st[10] + st[11]) / per
print(f"B={B}: eight-wave forward scan, 32-row phases x {np_f} per step; cycles per phase of workgroup 0 wave 0; total {v.sum():.0f} = {v.sum() / 32:.0f} per row")
for nm, x in zip(names, v):
    print(f"  {nm:48s} {x:8.1f}")
print(f"  of own rows landed + counter: own rows (blocking look) {st[8] / per:.1f}, the rest waiting for the other waves")
print(f"  of 'tile request + ...': strips of two phases ago {st[11] / per:.1f}, early look at the next rows {st[9] / per:.1f}, gate-input wait {st[10] / per:.1f}, the rest: released counter + arming + requests")
print(f"  own rows not there at the counted wait: {st[12] / per:.4f} per phase; MFMA phases repeated (somebody's rows missing): {st[13] / per:.4f}")
