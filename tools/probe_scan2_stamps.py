"""Diagnostic: cycle shares of the second-generation wide scans' phases (lstm_scan2.hip).
Needs the stamps build: make -C ocrd_keraslm_amd/csrc stamps"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rows = int(os.environ.get("KL_SCAN2_ROWS", "0")) or (32 if B // 32 // 32 >= 2 and B % 1024 == 0 else 16)
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
lib.kl_test_scan2_stamps.restype = C.c_int
lib.kl_test_scan2_stamps(None, 1)
n = 4
for _ in range(n):
    lm.train_window(idx, ctx, idx, None)
torch.cuda.synchronize()
st = (C.c_ulonglong * 32)()
lib.kl_test_scan2_stamps(st, 0)
st = list(st)
np_f = B // rows // 32
per_f = n * T * L * np_f
names_f = ['loop top', 'tile wait+check', 'barrier 1', 'MFMA phase (+acc init, table mode)', 'transposes + gate-input wait + next pieces',
           'gate math + LDS staging', 'barrier 2', 'stores + rotate']
v = np.array(st[:len(names_f)], dtype=np.float64) / per_f
print(f"B={B}: forward scan, {rows}-row phases x {np_f} per step; cycles per phase of workgroup 0 thread 0; total {v.sum():.0f} = {v.sum() / rows:.0f} per row")
for nm, x in zip(names_f, v):
    print(f"  {nm:44s} {x:8.1f}")
print(f"  (of 'tile wait+check': the counted wait alone {st[8] / per_f:.1f})")
print(f"  tiles that were requested too early: {st[12] / per_f:.4f} per phase; gate-input pieces not landed at the counted wait: {st[13] / per_f:.4f}")
np_b = B // 16 // 32
per_b = n * T * L * np_b
names_b = ['loop top + input loads', 'tile wait+check', 'barrier 1', 'MFMA + partial tiles', 'barrier 2', 'input wait', 'epilogue math + LDS',
           'barrier 3', 'publish + re-arm + rotate']
vb = np.array(st[16:25], dtype=np.float64) / per_b
print(f"B={B}: backward scan, 16-row blocks x {np_b} per step; total {vb.sum():.0f} = {vb.sum() / 16:.0f} per row")
for nm, x in zip(names_b, vb):
    print(f"  {nm:44s} {x:8.1f}")
print(f"  epilogue inputs not landed at the optimistic wait: {st[29] / per_b:.4f} per block")
print(f"  tiles that were requested too early: {st[28] / per_b:.4f} per block")
