"""Rater host logic vs golden fixtures produced by the REFERENCE's own code
(tests/golden/make_golden.py: reference Rater + stub model answered by the oracle).

CPU variant: our Rater over the oracle-backed test double -> must reproduce the
reference outputs exactly (strings, paths, call shapes) and to 1e-9 (scores).
GPU variant (-m gpu): our Rater over the HIP engine -> same strings / chosen
paths, probabilities within 1e-3 (north_star), scores to matching tolerance."""
import json
import os

import numpy as np
import pytest

from oracle import lstm_oracle as O
from ocrd_keraslm_amd.lib import Rater, Node
from ocrd_keraslm_amd.lib import windows
from tests.oracle_engine import OracleLM

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEAM = json.load(open(os.path.join(GOLD, "rater_seam.json")))
WINDOWS = json.load(open(os.path.join(GOLD, "windows.json")))


def hip_factory(*args):
    from ocrd_keraslm_amd.lib.engine import HipLM
    return HipLM(*args)


ENGINES = [pytest.param(OracleLM, 1e-9, id="oracle-cpu"),
           pytest.param(hip_factory, 1e-3, id="hip", marks=pytest.mark.gpu)]


def make_rater(factory, stateful, incremental):
    m = SEAM["model"]
    chars = m["chars"]
    r = Rater(engine_factory=factory)
    r.width, r.depth, r.length = m["width"], m["depth"], m["length"]
    r.stateful, r.incremental = stateful, incremental
    r.mapping = (dict((c, i) for i, c in enumerate(chars, 1)), dict((i, c) for i, c in enumerate(chars, 1)))
    r.voc_size = len(chars) + 1
    r.configure()
    cfg = O.ModelConfig(m["depth"], m["width"], r.voc_size, 1)
    w = O.init_weights(cfg, seed=m["seed"], emb_std=m["emb_std"], dtype=np.float64)
    r.model.set_weights(w, 3)
    r.status = 2
    if incremental:
        r.batch_size = 128
    return r


@pytest.mark.parametrize("case", WINDOWS, ids=lambda c: "T%d-n%d" % (c["length"], c["size"]))
def test_windows_match_reference(case):
    """rating.py:1005-1158 stateful windows incl. the zero-padded tail"""
    chars = SEAM["model"]["chars"]
    c_i = dict((c, i) for i, c in enumerate(chars, 1))
    got = list(windows.stateful_windows(case["text"], case["context"], case["length"], c_i))
    assert len(got) == len(case["batches"])
    assert len(got) == windows.count_windows(case["size"], case["length"])
    for (x, z, y), ref in zip(got, case["batches"]):
        assert x.tolist() == ref["x"]
        assert z[:, 0].tolist() == ref["ctx"]
        assert y.tolist() == ref["y"]


def test_node_ordering_quirks():
    class Alt:
        Unicode, conf, index = "bcd", 1.0, 0
    g = SEAM["node"]
    a = Node(state=None, value="a", cost=1.0)
    b = Node(state=None, value="b", cost=0.5, parent=a, extras=(None, Alt))
    c = Node(state=None, value="bc", cost=1.0, parent=a, extras=(None, Alt))
    assert b.pro_cost() == g["b_pro"] and c.pro_cost() == g["c_pro"]
    assert (b == c) == g["b_eq_c"] and (b < c) == g["b_lt_c"]
    assert len(c.to_sequence()) == g["seq_len"]
    assert (c in [b]) == g["c_in_list_of_b"]


@pytest.mark.parametrize("factory,tol", ENGINES)
def test_rate_matches_reference(factory, tol):
    """rating.py:493-529: consecutive rate() calls carry state (no reset)."""
    r = make_rater(factory, True, False)
    r.model.reset_states(1)
    for case in SEAM["rate"]:
        probs = r.rate(case["text"], case["context"])
        if case["probs"] is None:
            assert [float(p) for p in probs] == [1.0]
            continue
        assert len(probs) == len(case["probs"]) == len(case["text"])
        assert np.abs(np.array(probs, dtype=np.float64) - np.array(case["probs"])).max() < tol


@pytest.mark.parametrize("factory,tol", ENGINES)
def test_rate2_matches_reference(factory, tol):
    """rating.py:531-576 (the north_star's rate_once)"""
    r = make_rater(factory, True, False)
    for case in SEAM["rate2"]:
        res, ppl = r.rate_once(case["text"], case["context"])
        assert [c for c, _ in res] == [c for c, _ in case["result"]]
        assert np.abs(np.array([p for _, p in res]) - np.array([p for _, p in case["result"]])).max() < tol
        assert abs(ppl - case["perplexity"]) < max(tol * 1e3, 1e-6) * case["perplexity"]


@pytest.mark.parametrize("factory,tol", ENGINES)
def test_predict_matches_reference(factory, tol):
    """rating.py:578-639: list-of-arrays states in and out"""
    r = make_rater(factory, False, True)
    g = SEAM["predict"]
    preds, states = r.predict(list(g["candidates"][0]), [None, None, None], g["context"])
    assert len(states) == 3 and len(states[0]) == 4 and states[0][0].shape == (1, 32)
    preds2, states2 = r.predict(list(g["candidates"][1]), states, g["context"])
    assert np.abs(np.array(preds2) - np.array(g["preds2"])).max() < tol
    assert np.abs(np.array(states2) - np.array(g["states2"])).max() < max(tol, 1e-6) * 10


@pytest.mark.parametrize("factory,tol", ENGINES)
def test_generate_matches_reference(factory, tol):
    """rating.py:642-709: sampled text and the shape of every predict call"""
    r = make_rater(factory, False, True)
    for case in SEAM["generate"]:
        if hasattr(r.model, "step_calls"):
            r.model.step_calls.clear()
        out = r.generate(case["prefix"], case["length"], case["context"], case["variants"])
        assert out == case["result"]
        if hasattr(r.model, "step_calls"):
            assert r.model.step_calls == case["calls"]


class Alt(object):
    def __init__(self, text, conf, index):
        self.Unicode, self.conf, self.index = text, conf, index


class Elem(object):
    def __init__(self, id_):
        self.id = id_


def lattice(segments):
    import networkx as nx
    g = nx.DiGraph()
    for i, alts in enumerate(segments):
        g.add_edge(i, i + 1, element=Elem("e%d" % i), alternatives=[Alt(t, c, k) for k, (t, c) in enumerate(alts)])
    return g, 0, len(segments)


@pytest.mark.parametrize("factory,tol", ENGINES)
def test_rate_best_matches_reference(factory, tol):
    """rating.py:712-916: chosen beam path, scores, entropy, surviving beam and the
    batch sizes of all predict calls, over consecutive pages with carried traceback."""
    r = make_rater(factory, False, True)
    for case in SEAM["rate_best"]:
        traceback = None
        pages = []
        for segs in SEAM["lattices"]:
            g, s, e = lattice(segs)
            if hasattr(r.model, "step_calls"):
                r.model.step_calls.clear()
            path, entropy, traceback = r.rate_best(g, s, e, start_traceback=traceback, context=[17],
                                                   lm_weight=case["lm_weight"], beam_width=case["beam_width"],
                                                   beam_clustering_dist=case["dist"])
            pages.append((path, entropy, traceback, list(getattr(r.model, "step_calls", []))))
        path, entropy, traceback = r.next_path(traceback[0], ([], traceback[1]))
        pages.append((path, entropy, traceback, []))
        for (path, entropy, tb, calls), ref in zip(pages, case["pages"]):
            assert [[el.id, alt.Unicode] for el, alt, _ in path] == [[a, b] for a, b, _ in ref["path"]]
            scores = np.array([s for _, _, s in path])
            assert np.abs(scores - np.array([s for _, _, s in ref["path"]])).max(initial=0) < max(tol, 1e-9)
            assert abs(entropy - ref["entropy"]) < max(tol * 100, 1e-8)
            assert len(tb[0]) == len(ref["beam"])
            assert np.abs(np.array([n.cum_cost for n in tb[0]]) - np.array(ref["beam"])).max(initial=0) < max(tol * 100, 1e-8)
            if hasattr(r.model, "step_calls") and ref["calls"]:
                assert calls == ref["calls"]


@pytest.mark.parametrize("factory,tol", ENGINES)
def test_rate_best_exact_ties_match_reference(factory, tol):
    """Lattices whose alternatives (and therefore hypotheses) cost EXACTLY the same: which duplicate survives in the
    path and in the beam is decided by insort_left's tie rule and the order of the list operations alone."""
    r = make_rater(factory, False, True)
    for case in SEAM["rate_best_ties"]:
        traceback = None
        pages = []
        for segs in SEAM["tie_lattices"]:
            g, s, e = lattice(segs)
            if hasattr(r.model, "step_calls"):
                r.model.step_calls.clear()
            path, entropy, traceback = r.rate_best(g, s, e, start_traceback=traceback, context=[17],
                                                   lm_weight=case["lm_weight"], beam_width=case["beam_width"],
                                                   beam_clustering_dist=case["dist"])
            pages.append((path, entropy, traceback, list(getattr(r.model, "step_calls", []))))
        path, entropy, traceback = r.next_path(traceback[0], ([], traceback[1]))
        pages.append((path, entropy, traceback, []))
        exact = tol < 1e-6      # (on the GPU engine duplicates are still bitwise equal -- same inputs, same kernel -- but
        #                          near-ties between different texts may order differently: compare what is pinned)
        for (path, entropy, tb, calls), ref in zip(pages, case["pages"]):
            assert [[el.id, alt.Unicode] for el, alt, _ in path] == [[a, b] for a, b, _, _ in ref["path"]]
            if exact:
                assert [alt.index for _, alt, _ in path] == [i for _, _, i, _ in ref["path"]]
                assert [n.extras[1].index if n.extras else -1 for n in tb[0]] == [i for _, i in ref["beam"]]
                if hasattr(r.model, "step_calls") and ref["calls"]:
                    assert calls == ref["calls"]
            assert len(tb[0]) == len(ref["beam"])
            assert abs(entropy - ref["entropy"]) < max(tol * 100, 1e-8)


def test_status_machine_and_assertions():
    r = Rater(engine_factory=OracleLM)
    with pytest.raises(AssertionError):
        r.rate("abc")
    with pytest.raises(AssertionError):
        r.save("/tmp/x.npz")
    r = make_rater(OracleLM, True, False)
    with pytest.raises(AssertionError):
        r.predict(["a"], [None])          # stateful model cannot predict incrementally
    r = make_rater(OracleLM, False, True)
    with pytest.raises(AssertionError):
        r.rate("abc")
    with pytest.raises(AssertionError):
        r.predict(["a", "b"], [None])     # inconsistent lengths


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ocrd_keraslm_amd.lib import hipabi
    r = Rater()
    r.width, r.depth, r.length, r.voc_size = 32, 1, 8, 5
    with pytest.raises(hipabi.KlError):
        r.configure()
