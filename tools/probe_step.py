"""Where does a forward cell step spend its time?  Launches the raw step kernel with
degenerate strides to separate weight streaming, state-row reads and fixed cost."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
lib = hipabi.load()
lib.kl_test_fwd_step.restype = C.c_int
lib.kl_test_fwd_step.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
W, K = 512, 512
def run(B, lda, ldw, n_ops, fused, iters=200):
    A = torch.randn(max(B, 1) * K, device='cuda').to(torch.bfloat16)
    WT = torch.randn(4 * 4 * W * K, device='cuda').to(torch.bfloat16)
    c = torch.zeros(4 * B * W, device='cuda'); h = torch.zeros(4 * B * W, device='cuda', dtype=torch.bfloat16)
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(2):
        hipabi.check(lib.kl_test_fwd_step(p(A), lda, p(WT), ldw, K, n_ops, B, W, fused, 20, p(c), p(h), None))
    torch.cuda.synchronize(); t = time.time()
    hipabi.check(lib.kl_test_fwd_step(p(A), lda, p(WT), ldw, K, n_ops, B, W, fused, iters, p(c), p(h), None))
    torch.cuda.synchronize()
    return (time.time() - t) / iters * 1e6
for B in (1, 64, 256):
    for (lda, ldw, tag) in ((K, K, 'real'), (K, 0, 'W bcast'), (0, K, 'A bcast'), (0, 0, 'both bcast')):
        for n_ops, fused in ((1, 1), (2, 2)):
            print(f"B={B:4d} {tag:10s} ops={n_ops} fused={fused}: {run(B, lda, ldw, n_ops, fused):7.2f} us/launch", flush=True)
