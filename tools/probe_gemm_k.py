"""Diagnostic: the many-row GEMM (M = B*256, N = 2048, f32 out) at several K: separates main-loop from epilogue time."""
import sys
import torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
lib = hipabi.load()
M, N = 131072, 2048
s = torch.cuda.current_stream().cuda_stream
C = torch.zeros((M, N), device='cuda', dtype=torch.float32)
for K in (64, 128, 256, 512, 1024):
    A = (torch.rand((M, K), device='cuda') - 0.5).to(torch.bfloat16)
    B = (torch.rand((N, K), device='cuda') - 0.5).to(torch.bfloat16)
    def run():
        assert lib.kl_test_gemm_tn(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, K, K, N, 0, 1, s) == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"K={K:5d}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s  out {M*N*4/ms/1e6:6.0f} GB/s")
