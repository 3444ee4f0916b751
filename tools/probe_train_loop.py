"""Diagnostic: chars/s of Rater.train (host window generation + H2D + engine) against the engine-only bench."""
import cProfile
import io
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import Rater

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chars = "abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ.,;\n-"
rng = np.random.default_rng(0)
files = []
n_files = int(streams * 1.3) + 2
for k in range(n_files):
    f = io.StringIO(''.join(rng.choice(list(chars), 41000)))
    f.name = "anon_t%d_%d.txt" % (k, 1700 + k % 200)
    files.append(f)
r = Rater()
r.width, r.depth, r.length = 512, 2, 256
r.streams = streams
r.max_epochs = 1
r.seed = 1
r.configure()
import os
profile = os.environ.get('KL_PROBE_PROFILE', '1') != '0'     # (the profiler itself costs ~30 % here)
pr = cProfile.Profile()
t0 = time.time()
if profile:
    pr.enable()
r.train(files)
if profile:
    pr.disable()
dt = time.time() - t0
n_train = len(files) - int(np.ceil(len(files) * 0.2))
steps = max(1, int(np.ceil(n_train * int(np.ceil((41000 - 256) / 256)) / streams)))
print(f"streams={streams}: 1 epoch ({steps} steps of {streams}x256 chars + validation) in {dt:.2f} s "
      f"-> >= {steps * streams * 256 / dt / 1e6:.2f} Mchars/s incl. validation; history {r.history}")
if profile:
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
