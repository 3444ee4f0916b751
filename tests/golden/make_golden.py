"""Generate golden fixtures from the REFERENCE's own host logic.

Run here (the reference cannot travel to the GPU box; only its outputs are
committed):

    PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 \
        /opt/conda/bin/python3.9 tests/golden/make_golden.py

The reference's `ocrd_keraslm.lib.Rater` is imported unmodified.  Keras/TF are
not installed, so `Rater.model` is a stub object that answers the Keras calls the
reference makes (predict_on_batch, predict_generator, evaluate_generator,
reset_states, inputs) with the numpy oracle (oracle/lstm_oracle.py, float64)
on fixed seeded weights.  Everything ABOVE that seam -- windowing, padding,
probability extraction, beam search, lattice decoding, Node ordering, traceback
cutting -- is the reference's code, and its outputs are what the fixtures pin.

Outputs (tests/golden/): windows.json, rater_seam.json, stateful_train.json (the stateful training front end:
file-wise split, epoch sizes, window stream of a file list with resets and augmentations)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ocrd_keraslm.lib import Rater, Node   # the reference
import networkx as nx

from oracle import lstm_oracle as O

TEXT = ("Die Aufklaerung ist der Ausgang des Menschen aus seiner selbst verschuldeten Unmuendigkeit. "
        "Unmuendigkeit ist das Unvermoegen, sich seines Verstandes ohne Leitung eines anderen zu bedienen.\n"
        "Selbstverschuldet ist diese Unmuendigkeit, wenn die Ursache derselben nicht am Mangel des Verstandes, "
        "sondern der Entschliessung und des Muthes liegt, sich seiner ohne Leitung eines andern zu bedienen. "
        "Sapere aude! Habe Muth, dich deines eigenen Verstandes zu bedienen! ist also der Wahlspruch der Aufklaerung.\n")


class Named(object):
    def __init__(self, name):
        self.name = name


class StubModel(object):
    """Answers the Keras model calls of rating.py with the oracle."""

    def __init__(self, cfg, weights, stateful, incremental):
        self.cfg, self.w = cfg, weights
        self.stateful, self.incremental = stateful, incremental
        self.inputs = [Named('char_input')] + [Named('context%d_input' % (n + 1)) for n in range(cfg.n_ctx)]
        if incremental:
            for l in range(cfg.depth):
                self.inputs += [Named('initial_h_%d_input' % (l + 1)), Named('initial_c_%d_input' % (l + 1))]
        self.states = O.zero_states(cfg, 1, np.float64)
        self.calls = []

    def reset_states(self):
        self.states = O.zero_states(self.cfg, 1, np.float64)

    def _window(self, inputs):
        x = np.asarray(inputs[0]).astype(np.int64)
        ctx = np.stack([np.asarray(z).astype(np.int64) for z in inputs[1:1 + self.cfg.n_ctx]], axis=-1)
        probs, self.states, _ = O.forward_window(self.cfg, self.w, x, ctx, self.states)
        return probs

    def predict_on_batch(self, inputs):
        if self.incremental:
            x = np.asarray(inputs[0]).astype(np.int64)[:, 0]
            ctx = np.stack([np.asarray(z).astype(np.int64)[:, 0] for z in inputs[1:1 + self.cfg.n_ctx]], axis=-1)
            states = [np.asarray(s, dtype=np.float64) for s in inputs[1 + self.cfg.n_ctx:]]
            self.calls.append(len(x))
            probs, new = O.step_batch(self.cfg, self.w, x, ctx, states)
            return [probs] + new
        return self._window(inputs)

    def predict_generator(self, gen, steps, verbose=0):
        out = []
        for _ in range(steps):
            x, _y = next(gen)
            out.append(self._window(x))
        return np.concatenate(out, axis=0)

    def evaluate_generator(self, gen, steps, verbose=0):
        losses, accs = [], []
        for _ in range(steps):
            x, y = next(gen)
            probs = self._window(x)
            tgt = np.where(y.any(axis=-1), y.argmax(axis=-1), -1)
            ce, acc, _ = O.crossentropy(probs, tgt)
            losses.append(ce)
            accs.append(acc)
        return float(np.mean(losses)), float(np.mean(accs))


def make_rater(depth, width, length, stateful, incremental, seed=4):
    chars = sorted(set(TEXT))
    r = Rater()
    r.width, r.depth, r.length = width, depth, length
    r.stateful, r.incremental = stateful, incremental
    r.mapping = (dict((c, i) for i, c in enumerate(chars, 1)), dict((i, c) for i, c in enumerate(chars, 1)))
    r.voc_size = len(chars) + 1
    if stateful:
        r.variable_length = False
        r.first_window = 0
        r.batch_size = 1
    cfg = O.ModelConfig(depth, width, r.voc_size, 1)
    w = O.init_weights(cfg, seed=seed, emb_std=0.5, dtype=np.float64)
    r.model = StubModel(cfg, w, stateful, incremental)
    r.status = 2
    return r, cfg


class Alt(object):
    def __init__(self, text, conf, index):
        self.Unicode, self.conf, self.index = text, conf, index


class Elem(object):
    def __init__(self, id_):
        self.id = id_


def golden_windows():
    out = []
    for length, sizes in ((256, [1, 2, 100, 256, 257, 512, 513, 600]), (64, [63, 64, 65, 128, 129, 200]), (8, [3, 8, 9, 20])):
        r, _ = make_rater(1, 16, length, True, False)
        for size in sizes:
            text = (TEXT * 3)[:size]
            batches = []
            for x, y in r._gen_data(text, [173], length):
                tgt = np.where(y[0].any(axis=-1), y[0].argmax(axis=-1), -1)
                batches.append({"x": x[0][0].tolist(), "ctx": x[1][0].tolist(), "y": tgt.tolist()})
            out.append({"length": length, "size": size, "text": text, "context": [173], "batches": batches})
    return out


def lattice(segments):
    """linear lattice: node i --[alternatives]--> node i+1"""
    g = nx.DiGraph()
    for i, alts in enumerate(segments):
        g.add_edge(i, i + 1, element=Elem("e%d" % i),
                   alternatives=[Alt(t, c, k) for k, (t, c) in enumerate(alts)])
    return g, 0, len(segments)


LATTICES = [
    [[("Die", 0.9), ("Dle", 0.6), ("Dic", 0.5)], [(" ", 1.0)], [("Aufklaerung", 0.7), ("Aufklaernng", 0.65), ("Anfklaerung", 0.4)],
     [(" ", 1.0)], [("ist", 0.8), ("isl", 0.7)], [(" ", 1.0)], [("der", 0.9), ("dcr", 0.3), ("den", 0.5), ("des", 0.4)]],
    [[("Habe", 0.8), ("Hahe", 0.75)], [(" ", 1.0)], [("Muth", 0.6), ("Mnth", 0.6), ("Muht", 0.55)], [(",", 0.9), (".", 0.4)],
     [(" ", 1.0)], [("dich", 0.7), ("dieh", 0.69)], [(" ", 1.0)], [("deines", 0.9), ("deincs", 0.2)]],
]


TIE_LATTICES = [
    [[("Die", 0.9), ("Die", 0.9), ("Dle", 0.9)], [(" ", 1.0), (" ", 1.0)], [("Muth", 0.6), ("Mnth", 0.6), ("Muth", 0.6)],
     [(" ", 1.0)], [("ist", 0.8), ("ist", 0.8), ("isl", 0.8), ("ist", 0.8)]],
    [[("Habe", 0.75), ("Habe", 0.75)], [(" ", 1.0)], [("dich", 0.7), ("dich", 0.7), ("dieh", 0.7)]],
]


def golden_seam():
    out = {}
    # --- stateful: rate, rate2, test ---------------------------------------------------
    r, cfg = make_rater(2, 32, 16, True, False)
    texts = [TEXT[:50], TEXT[50:83], "x", "Muth!", TEXT[:16], TEXT[:17], TEXT[:33] + "äQ"]
    rates = []
    for t in texts:      # consecutive calls: state is carried (rate never resets)
        rates.append({"text": t, "context": [17], "probs": [float(p) for p in r.rate(t, [17])]}
                     if len(t) > 1 else {"text": t, "context": [17], "probs": None})
    out["rate"] = rates
    rate2 = []
    for t in (TEXT[:40], "Habe Muth", "äbc"):
        res, ppl = r.rate2(t, [3])
        rate2.append({"text": t, "context": [3], "result": [[c, float(p)] for c, p in res], "perplexity": float(ppl)})
    out["rate2"] = rate2
    # --- incremental: predict, generate, rate_best -------------------------------------
    r, cfg = make_rater(2, 32, 16, False, True)
    r.batch_size = 128
    preds, states = r.predict(list("Dax"), [None, None, None], [5])
    preds2, states2 = r.predict(list("ieb"), states, [5])
    out["predict"] = {"candidates": ["Dax", "ieb"], "context": [5],
                      "preds2": [p.tolist() for p in preds2],
                      "states2": [[s.tolist() for s in st] for st in states2]}
    gens = []
    for prefix, n, variants, ctx in (("Die Auf", 12, 3, [0]), ("H", 6, 5, [17]), ("zu be", 20, 1, [190])):
        r.model.calls = []
        gens.append({"prefix": prefix, "length": n, "variants": variants, "context": ctx,
                     "result": r.generate(prefix, n, ctx, variants), "calls": list(r.model.calls)})
    out["generate"] = gens
    bests = []
    for lm_weight, beam_width, dist in ((0.5, 10, 0), (0.8, 3, 5), (0.3, 10, 5)):
        traceback = None
        pages = []
        for segs in LATTICES:      # consecutive pages carry the traceback (rate.py:263-290)
            g, s, e = lattice(segs)
            r.model.calls = []
            path, entropy, traceback = r.rate_best(g, s, e, start_traceback=traceback, context=[17],
                                                   lm_weight=lm_weight, beam_width=beam_width,
                                                   beam_clustering_dist=dist)
            pages.append({"path": [[el.id, alt.Unicode, float(score)] for el, alt, score in path],
                          "entropy": float(entropy), "beam": [float(n.cum_cost) for n in traceback[0]],
                          "calls": list(r.model.calls)})
        path, entropy, traceback = r.next_path(traceback[0], ([], traceback[1]))
        pages.append({"path": [[el.id, alt.Unicode, float(score)] for el, alt, score in path],
                      "entropy": float(entropy), "beam": [float(n.cum_cost) for n in traceback[0]], "calls": []})
        bests.append({"lm_weight": lm_weight, "beam_width": beam_width, "dist": dist, "pages": pages})
    out["rate_best"] = bests
    # --- exact cost ties: duplicated alternatives (same text, same confidence, different index) and duplicated
    # incoming hypotheses cost exactly the same at every character, so WHICH of them survives is decided by the
    # insertion order of insort_left alone (rating.py:703, 807, 849)
    ties = []
    for lm_weight, beam_width, dist in ((0.5, 10, 0), (0.5, 2, 0), (0.7, 4, 5)):
        traceback = None
        pages = []
        for segs in TIE_LATTICES:
            g, s, e = lattice(segs)
            r.model.calls = []
            path, entropy, traceback = r.rate_best(g, s, e, start_traceback=traceback, context=[17],
                                                   lm_weight=lm_weight, beam_width=beam_width,
                                                   beam_clustering_dist=dist)
            pages.append({"path": [[el.id, alt.Unicode, alt.index, float(score)] for el, alt, score in path],
                          "entropy": float(entropy),
                          "beam": [[float(n.cum_cost), n.extras[1].index if n.extras else -1] for n in traceback[0]],
                          "calls": list(r.model.calls)})
        path, entropy, traceback = r.next_path(traceback[0], ([], traceback[1]))
        pages.append({"path": [[el.id, alt.Unicode, alt.index, float(score)] for el, alt, score in path],
                      "entropy": float(entropy),
                      "beam": [[float(n.cum_cost), n.extras[1].index if n.extras else -1] for n in traceback[0]], "calls": []})
        ties.append({"lm_weight": lm_weight, "beam_width": beam_width, "dist": dist, "pages": pages})
    out["rate_best_ties"] = ties
    out["tie_lattices"] = TIE_LATTICES
    out["model"] = {"depth": 2, "width": 32, "length": 16, "seed": 4, "emb_std": 0.5,
                    "chars": sorted(set(TEXT))}
    out["lattices"] = LATTICES
    # --- Node ordering quirks ------------------------------------------------------------
    a = Node(state=None, value="a", cost=1.0)
    b = Node(state=None, value="b", cost=0.5, parent=a, extras=(None, Alt("bcd", 1.0, 0)))
    c = Node(state=None, value="bc", cost=1.0, parent=a, extras=(None, Alt("bcd", 1.0, 0)))
    out["node"] = {"b_pro": b.pro_cost(), "c_pro": c.pro_cost(), "b_eq_c": bool(b == c), "b_lt_c": bool(b < c),
                   "seq_len": len(c.to_sequence()), "c_in_list_of_b": bool(c in [b])}
    return out


class MemFile(object):
    """an open text file as run.py hands them to Rater.train (rating.py:72-80): .name, .read(), .seek()"""

    def __init__(self, name, text):
        import io
        self.name = name
        self._f = io.StringIO(text)

    def read(self):
        return self._f.read()

    def seek(self, pos):
        return self._f.seek(pos)


class ResetRecorder(object):
    """stands in for ResetStatesCallback (callbacks.py:36-69): _gen_data_from_files calls .reset(name)"""

    def __init__(self, events):
        self.events = events

    def reset(self, name):
        self.events.append({"reset": name})


TRAIN_FILES = [("goethe_faust_1808.txt", 301), ("kant_kritik_1781.txt", 517), ("anon.txt", 129), ("a_b_1995.txt", 260),
               ("x_y_1650.txt", 64), ("lessing_nathan_1779.txt", 190), ("zz_top_1700.txt", 66)]


def golden_stateful_training():
    """rating.py:317-350 (stateful _split_data), 977-1002 (_gen_data_from_files) and 1062-1077 (train=True
    augmentations) of the unmodified reference, under fixed seeds of `random` and `numpy.random`."""
    import random
    length = 64
    texts = {}
    for k, (name, size) in enumerate(TRAIN_FILES):
        texts[name] = ((TEXT[7 * k:] + TEXT) * 3)[:size]

    def new_rater():
        r = Rater()
        r.width, r.depth, r.length = 16, 1, length
        r.stateful, r.incremental = True, False
        r.variable_length, r.first_window, r.batch_size = False, 0, 1
        r.status = 1
        return r

    def files():
        return [MemFile(name, texts[name]) for name, _ in TRAIN_FILES]

    out = {"length": length, "files": [{"name": n, "text": texts[n]} for n, _ in TRAIN_FILES]}
    splits = []
    for seed, with_val in ((7, False), (8, False), (9, True)):
        r = new_rater()
        data = files()
        val = data[-2:] if with_val else None
        if with_val:
            data = data[:-2]
        random.seed(seed)
        tr, va, split, n_tr, n_va, total, steps = r._split_data(data, val)
        splits.append({"seed": seed, "with_val": with_val, "train": [f.name for f in tr], "val": [f.name for f in va],
                       "split_is_none": split is None, "training_epoch_size": int(n_tr), "validation_epoch_size": int(n_va),
                       "total_size": int(total), "steps": int(steps), "voc_size": int(r.voc_size),
                       "chars": [r.mapping[1][i] for i in range(1, r.voc_size)]})
    out["split_data"] = splits
    gens = []
    for seed, train, cd, xd, repeat_windows in ((5, True, 0.3, 0.4, 0), (6, True, 0.01, 0.1, 0), (5, False, 0.3, 0.4, 0),
                                                (11, True, 0.5, 0.5, 12)):
        r = new_rater()
        data = files()
        random.seed(3)
        r._split_data(data, None)          # (vocabulary; the shuffled order is part of the fixture)
        order = [f.name for f in data]
        r.char_degradation, r.context_degradation = cd, xd
        events = []
        r.reset_cb = ResetRecorder(events)
        np.random.seed(seed)
        gen = r._gen_data_from_files(data, length, train=train, repeat=repeat_windows > 0)
        n = 0
        for x, y in gen:
            tgt = np.where(y[0].any(axis=-1), y[0].argmax(axis=-1), -1)
            events.append({"x": x[0][0].tolist(), "ctx": x[1][0].tolist(), "y": tgt.tolist()})
            n += 1
            if repeat_windows and n >= 40 + repeat_windows:      # (repeat=True never ends by itself)
                break
        gens.append({"np_seed": seed, "train": train, "char_degradation": cd, "context_degradation": xd,
                     "repeat": repeat_windows > 0, "order": order, "events": events})
    out["gen_data_from_files"] = gens
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "stateful_train.json"), "w") as f:
        json.dump(golden_stateful_training(), f)
    with open(os.path.join(HERE, "windows.json"), "w") as f:
        json.dump(golden_windows(), f)
    with open(os.path.join(HERE, "rater_seam.json"), "w") as f:
        json.dump(golden_seam(), f)
    print("golden fixtures written")


# ---------------------------------------------------------------------------------------
# G3: a model file written by the reference's own Rater.save (rating.py:918-945) on top of
# a Keras-2.3-style save_weights layout produced with h5py.
def golden_model_file(path):
    import h5py

    class SavingStub(StubModel):
        def save_weights(self, filename):
            cfg, w = self.cfg, self.w
            layers = [("char_input", []), ("context1_input", []),
                      ("char_embedding", [("embeddings:0", w["E"])]),
                      ("context1_embedding", [("embeddings:0", w["Ctx0"])]),
                      ("concat_hidden_input", [])]
            for l in range(cfg.depth):
                name = "lstm_%d" % (l + 1)
                # TF uniquifies variable scopes on the second configure(): lstm_1/lstm_1_1/kernel:0
                scope = name + ("_1" if l == 0 else "")
                layers.append((name, [(scope + "/kernel:0", w["K%d" % l]), (scope + "/recurrent_kernel:0", w["U%d" % l]),
                                      (scope + "/bias:0", w["b%d" % l])]))
                if l > 0:
                    layers.append(("dropout_%d" % l, []))
            layers.append(("char_output", []))
            with h5py.File(filename, "w") as f:
                f.attrs["layer_names"] = np.array([n.encode("utf8") for n, _ in layers])
                f.attrs["backend"] = b"tensorflow"
                f.attrs["keras_version"] = b"2.3.1"
                for name, weights in layers:
                    g = f.create_group(name)
                    names = [(wn if "/" in wn else name + "/" + wn) for wn, _ in weights]
                    g.attrs["weight_names"] = np.array([n.encode("utf8") for n in names]) if names else np.zeros((0,), "S1")
                    for n, (_, val) in zip(names, weights):
                        g.create_dataset(n, data=np.asarray(val, dtype=np.float32))

    r, cfg = make_rater(2, 32, 16, True, False)
    r.model.__class__ = SavingStub
    r.history = {"loss": [3.5, 3.25], "val_loss": [3.4, 3.3], "accuracy": [0.1, 0.2], "val_accuracy": [0.1, 0.15]}
    r.save(path)


if __name__ == "__main__":
    golden_model_file(os.path.join(HERE, "ref_model.h5"))
    print("reference model file written")
