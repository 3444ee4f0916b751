"""Diagnostic: the training step at the other shapes of SURVEY.md 8(d) (cfg2 at 1 / 64 streams, cfg5 at 512 streams),
as bench.py's small_batch / cfg5 blocks time them.  Usage: python tools/probe_shapes.py [small] [cfg5]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

what = sys.argv[1:] or ["small", "cfg5"]
device = torch.device("cuda:0")
rng = np.random.default_rng(1)
p = 1.0 / (np.arange(1, bench.VOC) + 1.0)
p /= p.sum()
corpus = rng.choice(bench.VOC - 1, size=bench.CORPUS, p=p).astype(np.int32) + 1
if "small" in what:
    sizes = [int(x) for x in os.environ.get("KL_SMALL_B", "1,16,64").split(",")]
    for Bs, st in [(b, 200 if b == 1 else 100) for b in sizes]:
        leg, lm = bench.training_leg(device, bench.DEPTH, bench.WIDTH, bench.LENGTH, bench.N_CTX, Bs, st, 10,
                                     corpus[:Bs * (bench.CORPUS // 64)])
        del lm
        print("cfg2 B=%d: %.3f ms/step, %.0f chars/s" % (Bs, leg["ms_per_step"], leg["value"]))
if "cfg5" in what:
    st5 = int(os.environ.get("KL_PROBE_STEPS", "10"))
    leg, lm = bench.training_leg(device, 4, 1024, 512, 2, 512, st5, min(3, st5), corpus)
    print("cfg5 B=512: %.2f ms/step, %.0f chars/s, mfma %.3f" % (leg["ms_per_step"], leg["value"], leg["mfma_frac"]))
    print(json.dumps(leg))
# any other shape: KL_SHAPE="depth,width,length,streams[,steps]" python tools/probe_shapes.py shape
if "shape" in what:
    for spec in os.environ.get("KL_SHAPE", "2,128,256,1024").split(";"):
        v = [int(x) for x in spec.split(",")]
        dep, wid, length, Bs = v[:4]
        st = v[4] if len(v) > 4 else 20
        leg, lm = bench.training_leg(device, dep, wid, length, bench.N_CTX, Bs, st, 3, corpus[:max(Bs, 64) * (bench.CORPUS // 1024)])
        del lm
        print("shape depth=%d width=%d length=%d streams=%d: %.3f ms/step, %.0f chars/s, mfma %.4f" % (dep, wid, length, Bs, leg["ms_per_step"], leg["value"], leg["mfma_frac"]))
