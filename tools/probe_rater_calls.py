"""Diagnostic: wall time of the Rater's string-level calls on the HIP engine (cfg2 topology)."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import Rater

chars = "abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ.,;\n-"


def make(stateful, incremental):
    r = Rater()
    r.width, r.depth, r.length = 512, 2, 256
    r.stateful, r.incremental = stateful, incremental
    r.mapping = ({c: i + 1 for i, c in enumerate(chars)}, {i + 1: c for i, c in enumerate(chars)})
    r.voc_size = len(chars) + 1
    r.configure()
    r.model.init_weights(seed=3, emb_std=0.5)
    r.status = 2
    return r


rng = np.random.default_rng(0)
text = ''.join(rng.choice(list(chars), 2000))
r = make(True, False)
for name, fn in (("rate 2000 chars", lambda: r.rate(text, [179])), ("rate2 500 chars", lambda: r.rate2(text[:500], [179]))):
    fn()
    t0 = time.time()
    fn()
    print(f"{name}: {time.time() - t0:.3f} s")
g = make(False, True)
g.generate("Die ", 20, [179], 2)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
out = g.generate("Die ", 100, [179], 4)
pr.disable()
print(f"generate 100 chars x 4 variants: {time.time() - t0:.3f} s")
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
