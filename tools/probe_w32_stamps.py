"""Diagnostic: cycle shares of the width-1024 backward scan's blocks (lstm_scan_w32.hip), workgroup 0: thread 0 (a computing
wave) and thread 256 (a DMA wave).  Needs the stamps build: make -C ocrd_keraslm_amd/csrc stamps
  python tools/probe_w32_stamps.py [B]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L, W, V, T = 4, 1024, 256, 512
lm = HipLM(L, W, V, 2)
lm.init_weights(seed=1)
lm.prepare(1)
rng = np.random.default_rng(0)
idx = torch.from_numpy(rng.integers(1, V, (B, T)).astype(np.int32)).cuda()
ctx = torch.zeros((B, T, 2), dtype=torch.int32).cuda()
masks = lm.draw_dropout_masks_device(B)
for _ in range(2):
    lm.train_window(idx, ctx, idx, masks)
torch.cuda.synchronize()
lib.kl_test_w32_stamps.restype = C.c_int
lib.kl_test_w32_stamps(None, 1)
n = 2
for _ in range(n):
    lm.train_window(idx, ctx, idx, masks)
torch.cuda.synchronize()
st = (C.c_ulonglong * 32)()
lib.kl_test_w32_stamps(st, 0)
st = np.array(list(st), dtype=np.float64)
blocks = n * T * L * ((B // 16 + 7) // 8)
names = ['(publish +) loop top', 'epilogue loads issued (waves 0-3)', 'tile fetched now (nothing requested ahead)', 'barrier: tile there', 'contraction',
         'partial sums to LDS', 'barrier: partial sums there, tile free', 'epilogue + staging (0-3) / next tile requested (4-7)',
         'publish (0-3) / requests landed (4-7)', '- / look at the landed tile (4-7)', '', '', '', '', '', '']
print(f"B={B}: width-1024 backward scan; cycles per block of workgroup 0 (clock64): wave 0 (epilogue + publish) / wave 4 (tile transport)")
for k in range(10):
    print(f"  {names[k]:40s} {st[k] / blocks:9.0f} {st[16 + k] / blocks:9.0f}")
print(f"  {'total':40s} {st[:14].sum() / blocks:9.0f} {st[16:30].sum() / blocks:9.0f}")

# ---- the forward scan
lib.kl_test_w32f_stamps.restype = C.c_int
st = (C.c_ulonglong * 32)()
lib.kl_test_w32f_stamps(st, 0)
st = np.array(list(st), dtype=np.float64) / (1.0 + 2.0 / n)      # (the two warm-up windows counted too: never reset for this array)
fnames = ['(stores +) loop top', 'tile fetched now (nothing requested ahead)', 'barrier: tile there', 'next tile requested (4-7), next inputs asked for',
          'contraction + partial sums to LDS', 'barrier: partial sums', 'epilogue + staging (0-3) / next tile landed and checked (4-7)', 'stores (0-3)']
print(f"B={B}: width-1024 forward scan; cycles per block of workgroup 0: wave 0 / wave 4")
for k in range(8):
    print(f"  {fnames[k]:40s} {st[k] / blocks:9.0f} {st[16 + k] / blocks:9.0f}")
print(f"  {'total':40s} {st[:8].sum() / blocks:9.0f} {st[16:24].sum() / blocks:9.0f}")
