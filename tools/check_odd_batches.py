"""Diagnostic: gradient parity against the f64 oracle at odd and very large batch sizes (shapes that leave the
wide scans' fast paths: no prefetch, thin scans, launch-per-step fallback).  Run on a GPU box from the repo root."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as t   # noqa: E402

for shape in [(2, 512, 64, 1001, 3, 1, True), (2, 512, 64, 1000, 3, 1, True), (2, 256, 40, 2056, 2, 1, False),
              (2, 512, 64, 3072, 2, 1, True)]:
    t.check_train_window_gradients(*shape)
    print("ok", shape, flush=True)
