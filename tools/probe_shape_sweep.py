"""Diagnostic: training throughput over model / batch shapes around cfg2, to find shapes that fall off the fast kernels.
  python tools/probe_shape_sweep.py            (depth, width, length, contexts, streams per line)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

device = torch.device("cuda:0")
rng = np.random.default_rng(1)
p = 1.0 / (np.arange(1, bench.VOC) + 1.0)
p /= p.sum()
corpus = rng.choice(bench.VOC - 1, size=bench.CORPUS, p=p).astype(np.int32) + 1
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (2, 512, 256, 1, 3072), (2, 512, 256, 2, 3072), (2, 512, 256, 0, 3072), (3, 512, 256, 1, 3072), (4, 512, 256, 1, 2048),
    (2, 512, 128, 1, 3072), (2, 512, 512, 1, 1024), (2, 512, 64, 1, 3072), (2, 500, 256, 1, 3072),
    (2, 256, 256, 1, 3072), (2, 256, 256, 1, 4096), (2, 128, 256, 1, 4096), (2, 1024, 256, 1, 512), (2, 1024, 256, 1, 1024),
    (2, 512, 256, 1, 256), (2, 512, 256, 1, 512), (2, 512, 256, 1, 128)]
for depth, width, length, n_ctx, B in shapes:
    try:
        leg, lm = bench.training_leg(device, depth, width, length, n_ctx, B, int(os.environ.get("KL_SWEEP_STEPS", "8")), 3, corpus)
        del lm
        torch.cuda.empty_cache()
        print("depth %d width %4d length %3d contexts %d streams %4d: %8.2f ms/step %7.2f M chars/s  %.1f %% of the MFMA roof"
              % (depth, width, length, n_ctx, B, leg["ms_per_step"], leg["value"] / 1e6, 100 * leg["mfma_frac"]), flush=True)
    except Exception as err:
        print("depth %d width %d length %d contexts %d streams %d: %r" % (depth, width, length, n_ctx, B, err), flush=True)
