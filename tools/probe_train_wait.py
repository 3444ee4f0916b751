"""Diagnostic: inside Rater.train, how long does read_loss wait (GPU not yet done) against the time the host
spends between two read_loss calls?"""
import io, sys, time
import numpy as np
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import Rater
from ocrd_keraslm_amd.lib.engine import HipLM
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chars = "abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ.,;\n-"
rng = np.random.default_rng(0)
files = []
for k in range(int(streams * 1.3) + 2):
    f = io.StringIO(''.join(rng.choice(list(chars), 41000)))
    f.name = "anon_t%d_%d.txt" % (k, 1700 + k % 200)
    files.append(f)
waits, gaps, last = [], [], [None]
orig = HipLM.read_loss
def timed(self, reset=True):
    t0 = time.perf_counter()
    if last[0] is not None:
        gaps.append(t0 - last[0])
    out = orig(self, reset)
    t1 = time.perf_counter()
    waits.append(t1 - t0)
    last[0] = t1
    return out
HipLM.read_loss = timed
for name in ("train_window", "draw_dropout_masks", "adam_step"):
    f0 = getattr(HipLM, name)
    acc = []
    def make(f0, acc):
        def w(self, *a, **k):
            t0 = time.perf_counter(); out = f0(self, *a, **k); acc.append(time.perf_counter() - t0); return out
        return w
    setattr(HipLM, name, make(f0, acc))
    globals()["acc_" + name] = acc
r = Rater()
r.width, r.depth, r.length = 512, 2, 256
r.streams = streams
r.max_epochs = 1
r.seed = 1
r.configure()
t0 = time.time()
r.train(files)
print(f"train() {time.time() - t0:.2f} s; read_loss calls {len(waits)}: wait mean {1e3 * np.mean(waits[5:160]):.2f} ms, "
      f"host time between calls mean {1e3 * np.mean(gaps[5:160]):.2f} ms; train_window enqueue {1e3 * np.mean(acc_train_window[5:]):.2f} ms, "
      f"masks {1e3 * np.mean(acc_draw_dropout_masks[5:]):.2f} ms, adam {1e3 * np.mean(acc_adam_step[5:]):.2f} ms")
