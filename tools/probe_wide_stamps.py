"""Diagnostic: cycle shares of the wide (64-unit) scans' step phases at B >= 512.
Needs the stamps build: make -C ocrd_keraslm_amd/csrc stamps"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
hipabi.LIB_PATH = os.path.join(os.path.dirname(hipabi.LIB_PATH), os.environ.get('KL_STAMPS_LIB', 'libkeraslm_hip_stamps.so'))
from ocrd_keraslm_amd.lib.engine import HipLM

lib = hipabi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
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
per = n * T * L * max(1, (B // 16) // 32)
v = np.array(list(st)[:11], dtype=np.float64) / per
names = ['loop top (zin loads)', 'poll', 'barrier1', 'tile load+LDS write', 'barrier2', 'MFMA+zt write', 'barrier3',
         'gate math+LDS pub', 'barrier4', 'store+vmcnt(0)', 'atomic+rest']
print(f"B={B}: wide forward scan, cycles (100 MHz clock64 ticks) per step of workgroup 0; total {v.sum():.0f}")
for nm, x in zip(names, v):
    print(f"  {nm:24s} {x:8.1f}")
print(f"  (wait for the gate inputs after barrier3: {st[11] / per:.1f}, taken out of 'gate math')")
print(f"  prefetches that came too early: {st[12] / per:.3f} per block")
vb = np.array(list(st)[16:27], dtype=np.float64) / per
print(f"B={B}: wide backward scan; total {vb.sum():.0f}")
for nm, x in zip(names, vb):
    print(f"  {nm:24s} {x:8.1f}")
print(f"  (wait for the epilogue inputs after barrier3: {st[27] / per:.1f}, taken out of 'gate math')")
print(f"  prefetches that came too early: {st[28] / per:.3f} per block")
