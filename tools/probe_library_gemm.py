"""Diagnostic: what the library GEMM (torch.matmul -> hipBLASLt / rocBLAS) does on the cfg2 training step's product shapes at 3072 streams,
for comparison with the hand-written kernels' times in profiles/r04_bench_B3072_kernel_stats.csv."""
import torch, time
K, M, N = 786432, 2048, 512
dZ = torch.randn(K, M, device='cuda', dtype=torch.bfloat16) * 0.01
H = torch.randn(K, N, device='cuda', dtype=torch.bfloat16)
def bench(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
fl = 2.0 * K * M * N
t1 = bench(lambda: torch.matmul(dZ.t(), H))
print(f"torch.matmul(dZ^T [2048 x K], H [K x 512]) bf16: {t1:.3f} ms = {fl / t1 / 1e9:.0f} TFLOP/s")
t2 = bench(lambda: torch.matmul(H.t(), dZ))
print(f"torch.matmul(H^T [512 x K], dZ [K x 2048]) bf16: {t2:.3f} ms = {fl / t2 / 1e9:.0f} TFLOP/s")
# dX = dZ [K x 2048] . Kn [2048 x 512]
Kn = torch.randn(M, N, device='cuda', dtype=torch.bfloat16)
t3 = bench(lambda: torch.matmul(dZ, Kn))
print(f"torch.matmul(dZ [K x 2048], K1 [2048 x 512]) bf16: {t3:.3f} ms = {fl / t3 / 1e9:.0f} TFLOP/s")
# P = X [K x 512] . KT [512 x 2048]
X = torch.randn(K, N, device='cuda', dtype=torch.bfloat16); KT = torch.randn(N, M, device='cuda', dtype=torch.bfloat16)
t4 = bench(lambda: torch.matmul(X, KT))
print(f"torch.matmul(X [K x 512], K^T [512 x 2048]) bf16: {t4:.3f} ms = {fl / t4 / 1e9:.0f} TFLOP/s")
