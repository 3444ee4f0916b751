"""The stateful training front end against fixtures generated from the UNMODIFIED reference
(tests/golden/make_golden.py: golden_stateful_training): file-wise train/validation split and epoch sizes
(rating.py:317-350), the window stream of a file list with its reset points (rating.py:977-1002) and the
train=True augmentations under fixed numpy seeds (rating.py:1062-1077).  VERDICT r1, item 5 (row a3)."""
import io
import json
import os
import random

import numpy as np
import pytest

from ocrd_keraslm_amd.lib import Rater, windows
from tests.oracle_engine import OracleLM

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "stateful_train.json")))


class MemFile(object):
    def __init__(self, name, text):
        self.name = name
        self._f = io.StringIO(text)

    def read(self):
        return self._f.read()

    def seek(self, pos):
        return self._f.seek(pos)


def files():
    return [MemFile(f["name"], f["text"]) for f in G["files"]]


def new_rater():
    r = Rater(engine_factory=OracleLM)
    r.width, r.depth, r.length = 16, 1, G["length"]
    r.stateful, r.incremental = True, False
    r.variable_length, r.first_window, r.batch_size = False, 0, 1
    r.status = 1
    return r


@pytest.mark.parametrize("case", G["split_data"], ids=lambda c: "seed%d%s" % (c["seed"], "-val" if c["with_val"] else ""))
def test_split_data_stateful_matches_reference(case):
    r = new_rater()
    data = files()
    val = data[-2:] if case["with_val"] else None
    if case["with_val"]:
        data = data[:-2]
    random.seed(case["seed"])
    tr, va, split, n_tr, n_va, total, steps = r._split_data(data, val)
    assert [f.name for f in tr] == case["train"]
    assert [f.name for f in va] == case["val"]
    assert (split is None) == case["split_is_none"]
    assert (int(n_tr), int(n_va), int(total), int(steps)) == (case["training_epoch_size"], case["validation_epoch_size"],
                                                              case["total_size"], case["steps"])
    assert r.voc_size == case["voc_size"]
    assert [r.mapping[1][i] for i in range(1, r.voc_size)] == case["chars"]


@pytest.mark.parametrize("case", G["gen_data_from_files"],
                         ids=lambda c: "np%d-%s-%g-%g%s" % (c["np_seed"], "train" if c["train"] else "eval", c["char_degradation"],
                                                             c["context_degradation"], "-repeat" if c["repeat"] else ""))
def test_file_windows_match_reference(case):
    r = new_rater()
    data = files()
    random.seed(3)
    r._split_data(data, None)
    assert [f.name for f in data] == case["order"]          # (the shuffle itself, in place as in the reference)
    events = []
    np.random.seed(case["np_seed"])
    n_expected = sum(1 for e in case["events"] if "x" in e)
    gen = windows.file_windows(data, r.length, r.mapping[0], train=case["train"], repeat=case["repeat"], rng=None,
                               on_new_file=(lambda name: events.append({"reset": name})) if case["train"] else None,
                               char_degradation=case["char_degradation"], context_degradation=case["context_degradation"])
    n = 0
    for x, z, y in gen:
        events.append({"x": x.tolist(), "ctx": z[:, 0].tolist(), "y": y.tolist()})
        n += 1
        if n >= n_expected:
            break
    # the reference passes ceil(year / 10) = 200 for years >= 1991 on to Embedding(200, 10), which is out of range
    # (latent bugs); this implementation clamps it to the last row
    want = []
    for e in case["events"]:
        if "x" in e:
            e = dict(e, ctx=[min(c, windows.CTX_VOCAB - 1) for c in e["ctx"]])
        want.append(e)
    resets_seen = [e for e in events if "reset" in e]
    resets_want = [e for e in want if "reset" in e]
    # (a generator that is stopped after its last expected window has not yet announced the next file)
    assert resets_seen[:len(resets_want)] == resets_want[:len(resets_seen)] and abs(len(resets_seen) - len(resets_want)) <= 1
    assert [e for e in events if "x" in e] == [e for e in want if "x" in e]
    # reset points sit at the same places of the window stream
    def positions(ev):
        out, k = [], 0
        for e in ev:
            if "reset" in e:
                out.append(k)
            else:
                k += 1
        return out
    m = min(len(resets_seen), len(resets_want))
    assert positions(events)[:m] == positions(want)[:m]
