"""Diagnostic: shader-clock stamps of inc_cell_kernel's phases (step_small.hip), one workgroup's wave 0.
Needs the stamps build: make -C ocrd_keraslm_amd/csrc stamps.   python tools/probe_inc_stamps.py [n]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
if not os.environ.get('KL_LIB'): hipabi.LIB_PATH = os.path.join(os.path.dirname(hipabi.LIB_PATH), 'libkeraslm_hip_stamps.so')
from ocrd_keraslm_amd.lib.engine import HipLM
lib = hipabi.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L, W, V = 2, 512, 256
lm = HipLM(L, W, V, 1)
lm.init_weights(seed=4, emb_std=0.5)
lm.prepare(3)
lm.ensure_pool(2 * n)
rng = np.random.default_rng(3)
ids = torch.from_numpy(rng.integers(1, V, size=(64, n)).astype(np.int32)).cuda()
cc = torch.from_numpy(rng.integers(0, 200, size=(n, 1)).astype(np.int32)).cuda()
a = torch.arange(n, dtype=torch.int32).cuda(); b = a + n
names = ['entry', 'indices + first weight half issued', 'rows requested (indices landed)', 'second half + epilogue inputs issued',
         'rows landed', 'rows in LDS', 'barrier', 'all loads landed', 'MFMAs done', 'partials exchanged', 'stores issued', 'stores acknowledged']
acc = np.zeros((2, 12)); cnt = 0
for s in range(64):
    lm.step_slots(ids[s], cc, a, b); a, b = b, a
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 32)()
    assert lib.kl_test_read_inc_stamps(st) == 0
    v = np.array(list(st), dtype=np.float64).reshape(2, 16)[:, :12]
    if s >= 8:
        acc += v - v[:, :1]; cnt += 1
acc /= cnt
for l in range(2):
    print(f"layer {l}: cumulative shader clocks (clock64) of workgroup (5, 3), wave 0, n = {n}:")
    for i, nm in enumerate(names):
        print(f"   {nm:40s} {acc[l, i]:9.0f}  (+{acc[l, i] - acc[l, i - 1] if i else 0:7.0f})")
