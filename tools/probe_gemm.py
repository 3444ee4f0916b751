"""Diagnostic: time kl_test_gemm_tn on the GEMM shapes of a cfg2 training window (B streams x 256 steps)."""
import sys

import torch

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi

lib = hipabi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
BT, W, V = B * 256, 512, 256
shapes = [  # name, M, N, K, out_mode, splits
    ("P      = H . K^T    (f32 out)", BT, 4 * W, W, 0, 1),
    ("dX     = dZ . Kn^T  (f32 out)", BT, W, 4 * W, 0, 1),
    ("logits = H . E^T    (f32 out)", BT, V, W, 0, 1),
    ("dH     = dl . ET^T  (f32 out)", BT, W, V, 0, 1),
    ("dU     = HT . dZT^T (atomic)", W, 4 * W, BT, 2, 8),
    ("dEKT   = dZT . OHT^T (atomic)", 4 * W, V, BT, 2, 8),
    ("dE     = dlT . HT^T (atomic)", V, W, BT, 2, 8),
]
only = sys.argv[2] if len(sys.argv) > 2 else None
s = torch.cuda.current_stream().cuda_stream
for name, M, N, K, mode, splits in shapes:
    if only and only not in name:
        continue
    A = (torch.rand((M, K), device='cuda') - 0.5).to(torch.bfloat16)
    Bm = (torch.rand((N, K), device='cuda') - 0.5).to(torch.bfloat16)
    Cm = torch.zeros((M, N), device='cuda', dtype=torch.float32)

    def run():
        rc = lib.kl_test_gemm_tn(A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(), None, M, N, K, K, K, N, mode, splits, s)
        assert rc == 0, rc
    run()
    torch.cuda.synchronize()
    # correctness on a sample of rows
    rows = torch.randint(0, M, (8,), device='cuda')
    ref = A[rows].float() @ Bm.float().t()
    err = (Cm[rows] - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
    Cm.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:32s} M={M:7d} N={N:5d} K={K:7d}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s  "
          f"out {M * N * 4 / ms / 1e6:6.0f} GB/s  relerr {err:.1e}")
