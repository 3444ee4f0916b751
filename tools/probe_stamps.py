"""Diagnostic: in-kernel stamps of the forward step kernel (needs `make -C ocrd_keraslm_amd/csrc stamps`)."""
import ctypes as C, sys, os
import numpy as np, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'ocrd_keraslm_amd', 'libkeraslm_hip_stamps.so'))
lib.kl_test_fwd_step.restype = C.c_int
lib.kl_test_fwd_step.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
lib.kl_zero_page_ready = getattr(lib, '_Z18kl_zero_page_readyv')
W = K = 512
names = ['entry', 'S copied', 'addresses', 'loads issued', 'loads landed', 'mfma done', 'barrier', 'gate math', 'stores issued']
for B in (1, 64):
    A = torch.randn(B * K, device='cuda').to(torch.bfloat16)
    WT = torch.randn(4 * 4 * W * K, device='cuda').to(torch.bfloat16)
    c = torch.zeros(4 * B * W, device='cuda'); h = torch.zeros(4 * B * W, device='cuda', dtype=torch.bfloat16)
    p = lambda t: C.c_void_p(t.data_ptr())
    acc = np.zeros(9)
    n = 0
    for it in range(30):
        assert lib.kl_test_fwd_step(p(A), K, p(WT), K, K, 2, B, W, 2, 1, p(c), p(h), None) == 0
        torch.cuda.synchronize()
        st = (C.c_ulonglong * 16)()
        assert lib.kl_test_read_stamps(st) == 0
        v = np.array(list(st)[:9], dtype=np.float64)
        if it >= 5:
            acc += v - v[0]; n += 1
    acc /= n
    print(f"B={B}: cumulative shader cycles (clock64):")
    for i, nm in enumerate(names):
        print(f"   {nm:14s} {acc[i]:9.0f}  (+{acc[i]-acc[i-1] if i else 0:7.0f})")
