"""Diagnostic: time the K-major weight-gradient contraction (kl_test_gemm_an) at the shapes of a cfg2 window."""
import sys
import torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
lib = hipabi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
BT, W, V = B * 256, 512, 256
s = torch.cuda.current_stream().cuda_stream
for name, M, N, K, b_km in [("dU^T = dZ^T . H", 4 * W, W, BT, 1), ("dE = dl^T . H", V, W, BT, 1), ("dEK^T = dZ^T . OH", 4 * W, V, BT, 0)]:
    A = (torch.rand((K, M), device='cuda') - 0.5).to(torch.bfloat16)
    Bm = (torch.rand((K, N) if b_km else (N, K), device='cuda') - 0.5).to(torch.bfloat16)
    C = torch.zeros((N, M), device='cuda', dtype=torch.float32)
    def run():
        rc = lib.kl_test_gemm_an(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, M, N if b_km else K, M, 1, b_km, s)
        assert rc == 0, rc
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:22s} M={M:5d} N={N:4d} K={K:7d}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s")
