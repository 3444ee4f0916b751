"""Golden fixtures for the STATELESS window mode, from the reference's own host logic.

    PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 \
        /opt/conda/bin/python3.9 tests/golden/make_golden_stateless.py

As in make_golden.py the reference's `Rater` is imported unmodified and `Rater.model` is a stub that answers
the Keras calls with the numpy oracle -- here with the semantics of the stateless graph (rating.py:93-99,
126-129): every window starts from zero state and only the LAST position's distribution comes back.
Pinned: `_gen_data` in stateless mode (prediction and training windows, partial windows, the three
augmentations, the shared train/validation split array), `_split_data` epoch sizes, `rate`, `rate2`, `test`.

Output: tests/golden/stateless.json
"""
import io
import json
import os
import random
import sys
from math import ceil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ocrd_keraslm.lib import Rater   # the reference

from oracle import lstm_oracle as O

TEXT = ("Die Aufklaerung ist der Ausgang des Menschen aus seiner selbst verschuldeten Unmuendigkeit. "
        "Unmuendigkeit ist das Unvermoegen, sich seines Verstandes ohne Leitung eines anderen zu bedienen.\n"
        "Sapere aude! Habe Muth, dich deines eigenen Verstandes zu bedienen!\n")
LENGTH, BATCH = 16, 8


class Named(object):
    def __init__(self, name):
        self.name = name


class StatelessStub(object):
    """Keras model stand-in for stateful=False, incremental=False: input [b, L'] -> probs of the last position [b, V]"""

    def __init__(self, cfg, weights):
        self.cfg, self.w = cfg, weights
        self.inputs = [Named('char_input')] + [Named('context%d_input' % (n + 1)) for n in range(cfg.n_ctx)]

    def _window(self, inputs):
        x = np.asarray(inputs[0]).astype(np.int64)
        ctx = np.stack([np.asarray(z).astype(np.int64) for z in inputs[1:1 + self.cfg.n_ctx]], axis=-1)
        probs, _, _ = O.forward_window(self.cfg, self.w, x, ctx, O.zero_states(self.cfg, x.shape[0], np.float64))
        return probs[:, -1, :]

    def predict_on_batch(self, inputs):
        return self._window(inputs)

    def predict_generator(self, gen, steps, verbose=0):
        out = []
        for _ in range(steps):
            x, _y = next(gen)
            out.append(self._window(x))
        return np.concatenate(out, axis=0)

    def evaluate_generator(self, gen, steps, verbose=0):
        # Keras averages the batch losses weighted by batch size
        losses, accs, sizes = [], [], []
        for _ in range(steps):
            x, y = next(gen)
            p = self._window(x)
            tgt = y.argmax(axis=-1)
            has = y.any(axis=-1)
            pc = np.clip(p[np.arange(len(tgt)), tgt], 1e-7, 1 - 1e-7)
            losses.append(float(np.mean(np.where(has, -np.log(pc), 0.0))))
            accs.append(float(np.mean(p.argmax(axis=-1) == np.where(has, tgt, 0))))
            sizes.append(len(tgt))
        return float(np.average(losses, weights=sizes)), float(np.average(accs, weights=sizes))


def make_rater(variable_length, seed=4):
    chars = sorted(set(TEXT))
    r = Rater()
    r.width, r.depth, r.length = 32, 2, LENGTH
    r.stateful, r.incremental = False, False
    r.variable_length = variable_length
    r.batch_size = BATCH
    r.mapping = (dict((c, i) for i, c in enumerate(chars, 1)), dict((i, c) for i, c in enumerate(chars, 1)))
    r.voc_size = len(chars) + 1
    cfg = O.ModelConfig(2, 32, r.voc_size, 1)
    w = O.init_weights(cfg, seed=seed, emb_std=0.5, dtype=np.float64)
    r.model = StatelessStub(cfg, w)
    r.status = 2
    return r


def dump_batches(gen, limit=400):
    out = []
    for k, (x, y) in enumerate(gen):
        if k >= limit:
            break
        y = np.asarray(y)
        tgt = np.where(y.any(axis=-1), y.argmax(axis=-1), -1)
        out.append({"x": np.asarray(x[0]).astype(int).tolist(), "ctx": np.asarray(x[1]).astype(int).tolist(),
                    "y": tgt.astype(int).tolist()})
    return out


def main():
    gold = {"text": TEXT, "length": LENGTH, "batch_size": BATCH, "gen": {}}
    text = TEXT[:75]
    # ---- _gen_data, prediction mode (steps 1)
    for vl in (True, False):
        r = make_rater(vl)
        gold["gen"]["predict_vl%d" % vl] = dump_batches(r._gen_data(text, [179], 1, False, None))
    # ---- _gen_data, training mode (steps 3), random numbers from the seeded global generator
    for vl in (True, False):
        for rates in ((0.01, 0.1, 0.1), (0.3, 0.3, 0.4)):
            r = make_rater(vl)
            r.char_degradation, r.context_degradation, r.first_window = rates
            np.random.seed(5)
            key = "train_vl%d_%s" % (vl, "default" if rates[0] == 0.01 else "boosted")
            gold["gen"][key] = {"rates": rates, "seed": 5, "batches": dump_batches(r._gen_data(TEXT, [179], 3, True, None))}
    # ---- shared split array: training and validation generators over the same text
    r = make_rater(True)
    r.char_degradation, r.context_degradation, r.first_window = 0.3, 0.3, 0.4
    np.random.seed(7)
    split = np.random.uniform(0, 1, (ceil(len(TEXT) / 3),))
    gold["gen"]["split"] = {"split": split.tolist(),
                            "train": dump_batches(r._gen_data(TEXT, [179], 3, True, split)),
                            "val": dump_batches(r._gen_data(TEXT, [179], 3, False, split))}
    # ---- _split_data (stateless branch): epoch sizes, steps, mapping, split array
    files = []
    for k, (name, a, b) in enumerate((("anon_t0_1784.txt", 0, 120), ("anon_t1_1801.txt", 100, 260))):
        f = io.StringIO(TEXT[a:b])
        f.name = name
        files.append(f)
    r = make_rater(True)
    r.mapping = ({}, {})
    r.status = 1
    random.seed(3)
    np.random.seed(11)
    tr, va, split, tsize, vsize, total, steps = r._split_data(list(files), None)
    gold["split_data"] = {"files": [[f.name, f.getvalue()] for f in files], "random_seed": 3, "np_seed": 11,
                          "training_epoch_size": float(tsize), "validation_epoch_size": float(vsize), "total_size": int(total),
                          "steps": int(steps), "split": split.tolist(), "order": [f.name for f in tr],
                          "chars": sorted(r.mapping[0].keys()), "voc_size": int(r.voc_size)}
    # ---- rate / rate2 / test on the stub model
    gold["rate"] = {}
    for vl in (True, False):
        r = make_rater(vl)
        probs = r.rate(TEXT[:60], [179])
        res, ppl = r.rate2(TEXT[:40], [179])
        f = io.StringIO(TEXT[:90])
        f.name = "anon_t0_1784.txt"
        pp = r.test([f])
        gold["rate"]["vl%d" % vl] = {"rate_text": TEXT[:60], "rate": [float(p) for p in probs],
                                     "rate2_text": TEXT[:40], "rate2": [[c, float(p)] for c, p in res], "rate2_ppl": float(ppl),
                                     "test_text": TEXT[:90], "test_name": f.name, "test_ppl": float(pp)}
    with open(os.path.join(HERE, "stateless.json"), "w") as out:
        json.dump(gold, out)
    print("wrote stateless.json:", {k: (len(v) if hasattr(v, '__len__') else v) for k, v in gold["gen"].items()})


if __name__ == "__main__":
    main()
