"""Stateless window mode against fixtures produced by the REFERENCE's own host logic
(tests/golden/make_golden_stateless.py: `_gen_data`, `_split_data`, `rate`, `rate2`, `test` of the unmodified
reference Rater with stateful=False, its Keras model replaced by an oracle-backed stub)."""
import io
import json
import os
import random
from math import ceil

import numpy as np
import pytest

from ocrd_keraslm_amd.lib import Rater, windows
from tests.oracle_engine import OracleLM

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "stateless.json")))
TEXT, LENGTH, BATCH = GOLD["text"], GOLD["length"], GOLD["batch_size"]
CHARS = sorted(set(TEXT))
C_I = dict((c, i) for i, c in enumerate(CHARS, 1))


def same_batches(mine, gold):
    mine = list(mine)
    assert len(mine) == len(gold), (len(mine), len(gold))
    for k, ((x, z, y), g) in enumerate(zip(mine, gold)):
        assert x.tolist() == g["x"], k
        assert z[:, :, 0].tolist() == g["ctx"], k
        assert y.tolist() == g["y"], k


@pytest.mark.parametrize("vl", [True, False])
def test_prediction_windows(vl):
    gen = windows.stateless_batches(TEXT[:75], [179], LENGTH, C_I, 1, batch_size=BATCH, train=False, variable_length=vl)
    same_batches(gen, GOLD["gen"]["predict_vl%d" % vl])


@pytest.mark.parametrize("vl", [True, False])
@pytest.mark.parametrize("which", ["default", "boosted"])
def test_training_windows_and_augmentations(vl, which):
    g = GOLD["gen"]["train_vl%d_%s" % (vl, which)]
    np.random.seed(g["seed"])
    gen = windows.stateless_batches(TEXT, [179], LENGTH, C_I, 3, batch_size=BATCH, train=True, variable_length=vl,
                                    char_degradation=g["rates"][0], context_degradation=g["rates"][1], first_window=g["rates"][2])
    same_batches(gen, g["batches"])


def test_shared_split_array():
    g = GOLD["gen"]["split"]
    split = np.asarray(g["split"])
    kw = dict(batch_size=BATCH, variable_length=True, char_degradation=0.3, context_degradation=0.3, first_window=0.4, split=split)
    same_batches(windows.stateless_batches(TEXT, [179], LENGTH, C_I, 3, train=True, **kw), g["train"])
    same_batches(windows.stateless_batches(TEXT, [179], LENGTH, C_I, 3, train=False, **kw), g["val"])


def hip_factory(*args):
    from ocrd_keraslm_amd.lib.engine import HipLM
    return HipLM(*args)


def make_rater(vl, factory=OracleLM):
    r = Rater(engine_factory=factory)
    r.width, r.depth, r.length = 32, 2, LENGTH
    r.stateful, r.incremental = False, False
    r.variable_length = vl
    r.batch_size = BATCH
    r.mapping = (dict(C_I), dict((i, c) for c, i in C_I.items()))
    r.voc_size = len(CHARS) + 1
    r.configure()
    r.model.init_weights(seed=4, emb_std=0.5)
    r.status = 2
    return r


def test_split_data_stateless():
    g = GOLD["split_data"]
    files = []
    for name, content in g["files"]:
        f = io.StringIO(content)
        f.name = name
        files.append(f)
    r = Rater(engine_factory=OracleLM)
    r.width, r.depth, r.length = 32, 2, LENGTH
    r.stateful, r.incremental, r.batch_size = False, False, BATCH
    r.status = 1
    random.seed(g["random_seed"])
    np.random.seed(g["np_seed"])
    tr, va, split, tsize, vsize, total, steps = r._split_data(list(files), None)
    assert [f.name for f in tr] == g["order"]
    assert (float(tsize), float(vsize), int(total), int(steps)) == (g["training_epoch_size"], g["validation_epoch_size"],
                                                                     g["total_size"], g["steps"])
    assert np.allclose(split, g["split"], atol=0)
    assert sorted(r.mapping[0].keys()) == g["chars"] and r.voc_size == g["voc_size"]


@pytest.mark.parametrize("factory,tol", [pytest.param(OracleLM, 1e-9, id="oracle-cpu"),
                                         pytest.param(hip_factory, 1e-3, id="hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("vl", [True, False])
def test_rate_rate2_test_stateless(vl, factory, tol):
    g = GOLD["rate"]["vl%d" % vl]
    r = make_rater(vl, factory)
    probs = r.rate(g["rate_text"], [179])
    # (the reference stops after ceil((size-1)/batch_size) generator batches, partial windows included:
    #  a reproduced quirk of its stateless scoring)
    assert len(probs) == len(g["rate"])
    assert np.abs(np.array(probs, dtype=np.float64) - np.array(g["rate"])).max() < tol
    res, ppl = r.rate2(g["rate2_text"], [179])
    assert [c for c, _ in res] == [c for c, _ in g["rate2"]]
    assert np.abs(np.array([p for _, p in res], dtype=np.float64) - np.array([p for _, p in g["rate2"]])).max() < tol
    assert abs(ppl - g["rate2_ppl"]) < 10 * tol * g["rate2_ppl"]
    f = io.StringIO(g["test_text"])
    f.name = g["test_name"]
    assert abs(r.test([f]) - g["test_ppl"]) < 10 * tol * g["test_ppl"]


def test_train_stateless_runs_on_the_test_double(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)      # (training writes its per-epoch checkpoints into the working directory)
    random.seed(1)
    np.random.seed(2)
    files = []
    for k in range(2):
        f = io.StringIO((TEXT * 2)[k * 40:k * 40 + 220])
        f.name = "anon_t%d_%d.txt" % (k, 1784 + k)
        files.append(f)
    r = Rater(engine_factory=OracleLM)
    r.width, r.depth, r.length = 12, 2, 8
    r.stateful, r.incremental, r.batch_size = False, False, 16
    r.max_epochs = 2
    r.configure()
    r.train(files)
    assert r.status == 2
    assert len(r.history["loss"]) >= 1 and all(np.isfinite(v) for v in r.history["loss"] + r.history["val_loss"])
    probs = r.rate(TEXT[:30], [179])
    assert probs[0] == 1.0 and all(0.0 <= p <= 1.0 for p in probs)
