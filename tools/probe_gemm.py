"""Diagnostic: time of the activation x weight contractions at the training step's shapes (kl_test_gemm_tn hook)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi

lib = hipabi.load()
lib.kl_test_gemm_tn.restype = C.c_int
lib.kl_test_gemm_tn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long,
                                C.c_int, C.c_int, C.c_void_p]


def run(M, N, K, out_mode, reps=5):
    a = (torch.randn(M, K, device='cuda') * 0.1).to(torch.bfloat16)
    b = (torch.randn(N, K, device='cuda') * 0.1).to(torch.bfloat16)
    c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16 if out_mode == 1 else torch.float32)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        assert lib.kl_test_gemm_tn(a.data_ptr(), b.data_ptr(), c.data_ptr(), None, M, N, K, K, K, N, out_mode, 1, s) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.kl_test_gemm_tn(a.data_ptr(), b.data_ptr(), c.data_ptr(), None, M, N, K, K, K, N, out_mode, 1, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ref = (a[:256].float() @ b.float().t())
    err = (c[:256].float() - ref).abs().max().item() / ref.abs().max().item()
    byt = M * K * 2 + N * K * 2 + M * N * (2 if out_mode == 1 else 4)
    print(f"M={M} N={N} K={K} out={'bf16' if out_mode == 1 else 'f32'}: {ms:.3f} ms, {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s, {byt / ms / 1e9:.2f} TB/s, err {err:.1e}", flush=True)


shapes = [tuple(int(x) for x in s.split(',')) for s in sys.argv[1:]] or [(262144, 2048, 512, 1), (786432, 2048, 512, 1)]
for M, N, K, om in shapes:
    run(M, N, K, om)
