"""Diagnostic: cycle shares of the persistent forward scan's step phases.
Needs the stamps build: make -C ocrd_keraslm_amd/csrc stamps"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
hipabi.LIB_PATH = os.path.join(os.path.dirname(hipabi.LIB_PATH), 'libkeraslm_hip_stamps.so')
from ocrd_keraslm_amd.lib.engine import HipLM

lib = hipabi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L, W, V, T = 2, 512, 256, 256
lm = HipLM(L, W, V, 1)
lm.init_weights(seed=1)
lm.prepare(1)
rng = np.random.default_rng(0)
idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
ctx = torch.zeros((B, T, 1), dtype=torch.int32).cuda()
for _ in range(3):
    lm.train_window(idx, ctx, idx, None)
torch.cuda.synchronize()
lib.kl_test_scan_stamps(None, 1)
n = 5
for _ in range(n):
    lm.train_window(idx, ctx, idx, None)
torch.cuda.synchronize()
st = (C.c_ulonglong * 32)()
lib.kl_test_scan_stamps(st, 0)
v = np.array(list(st)[:11], dtype=np.float64) / (n * T)
names = ['loop top', 'poll', 'barrier1', 'A load+LDS write', 'barrier2', 'MFMA+zt write', 'barrier3', 'gate math',
         'stores issue', 'vmcnt(0)', 'barrier4+atomic']
print(f"B={B}: forward scan, cycles per time step of the last workgroup (top layer); total {v.sum():.0f}")
for nm, x in zip(names, v):
    print(f"  {nm:18s} {x:8.0f}")
vb = np.array(list(st)[16:25], dtype=np.float64) / (n * T)
bnames = ['loop top (plain loads)', 'poll', 'barrier1', 'sc1 loads + MFMA + zt', 'barrier2', 'gate math', 'stores issue',
          'vmcnt(0)', 'barrier3+atomic']
print(f"B={B}: backward scan, workgroup 0; total {vb.sum():.0f}")
for nm, x in zip(bnames, vb):
    print(f"  {nm:24s} {x:8.0f}")
