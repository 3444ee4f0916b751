"""Comparison of the engine's gradient arrays with the f64 oracle's, shared by the GPU parity tests and
tools/fuzz_parity.py.

The loss is mean CE + embedding regularisers (rating.py:187-246).  For the embedding tables `E` / `Ctx*` the
regularisers' ANALYTIC part (-0.04 (1 - |E_r|^2) E_r, the row-0 terms ...) is 1e2 ... 1e5 times larger than what
back-propagation delivers (the tied output layer's `dlogits^T H`, rating.py:155-168, and the scatter of layer 0's
input gradient into the rows the window used, rating.py:103-125), so a comparison of the TOTAL gradient says nothing
about the latter.  Hence, for those arrays:

* the total is compared as for every other array (this is the check of the regulariser kernels), and
* the oracle's regulariser gradient is subtracted from BOTH sides in f64 and the remainder -- the back-propagated part --
  is held to the same relative-L2 / max-norm bounds as K / U / b against `backward_window(..., with_regularisers=False)`.
  Row 0 carries regulariser terms of several hundred times the row's own size (sums over all rows), so the f32 rounding
  of the engine's total alone is ~1e-3 of the back-propagated part there: row 0 gets an extra absolute allowance of a few
  f32 ulps of its regulariser part.  The same holds, at a smaller scale, for the other rows (the engine delivers ONE f32
  number per entry: what it can say about a back-propagated part of 1e-6 beside a regulariser part of 1 ends at 6e-8), so
  every bound on the back-propagated part carries an allowance of 3e-7 of the regulariser part beside it.  Tests that want
  the context tables' back-propagated gradient resolved therefore use SMALL context embeddings (`small_ctx_tables`: the
  regulariser gradients scale with the tables, the back-propagated part does not)."""
from collections import OrderedDict

import numpy as np

F32_ALLOWANCE = 3e-7      # a few f32 ulps (6e-8) of the regulariser part that shares the entry

_weights = OrderedDict()


def cached_weights(depth, width, voc, n_ctx=1, seed=4, emb_std=0.5):
    """`oracle.init_weights` costs 1.5-3 s at widths 512-1024 (orthogonal initialisers) and a few topologies serve most GPU
    tests: the arrays are kept (read-only, the eight most recent) instead of being drawn again per test."""
    from oracle import lstm_oracle as O
    key = (depth, width, voc, n_ctx, seed, float(emb_std))
    w = _weights.get(key)
    if w is None:
        w = O.init_weights(O.ModelConfig(depth, width, voc, n_ctx), seed=seed, emb_std=emb_std, dtype=np.float32)
        for v in w.values():
            v.setflags(write=False)
        _weights[key] = w
        while len(_weights) > 8:
            _weights.popitem(last=False)
    else:
        _weights.move_to_end(key)
    return dict(w)


def small_ctx_tables(w, factor=0.01):
    """the same weights with the context embeddings scaled down (see the module text)"""
    return {k: (v * np.float32(factor) if k.startswith("Ctx") else v) for k, v in w.items()}


def gradient_table(layout, grads, g_data, rg):
    """-> {name: {...}}: errors of every array.  `g_data`: the oracle's gradient WITHOUT regularisers (already scaled
    as the engine's mean), `rg`: `oracle.regulariser_grads` (only E / Ctx* present)."""
    table = {}
    for name, _off, _rows, _cols in layout:
        data = np.asarray(g_data[name], dtype=np.float64)
        got = np.asarray(grads[name], dtype=np.float64).reshape(data.shape)
        reg = rg.get(name)
        total = data if reg is None else data + reg
        e = {"rel_l2": float(np.linalg.norm(got - total) / (np.linalg.norm(total) + 1e-300)),
             "max_over_maxnorm": float(np.abs(got - total).max() / (np.abs(total).max() + 1e-300)),
             "ref_norm": float(np.linalg.norm(total))}
        if reg is not None:
            d = got - reg - data                        # error of the back-propagated part
            body, body_ref = d[1:], data[1:]
            e["data_rel_l2"] = float(np.linalg.norm(body) / (np.linalg.norm(body_ref) + 1e-300))
            e["data_max_over_maxnorm"] = float(np.abs(body).max() / (np.abs(body_ref).max() + 1e-300))
            e["data_ref_norm"] = float(np.linalg.norm(body_ref))
            e["data_err_norm"] = float(np.linalg.norm(body))
            e["data_err_max"] = float(np.abs(body).max())
            e["data_ref_max"] = float(np.abs(body_ref).max())
            e["reg_body_norm"] = float(np.linalg.norm(reg[1:]))
            e["reg_body_max"] = float(np.abs(reg[1:]).max())
            e["row0_err"] = float(np.linalg.norm(d[0]))
            e["row0_data_norm"] = float(np.linalg.norm(data[0]))
            e["row0_reg_norm"] = float(np.linalg.norm(reg[0]))
        table[name] = e
    return table


def assert_gradients(layout, grads, g_data, rg, rel=1.5e-2, maxn=3e-2, where=()):
    """Raises AssertionError naming the array and the measure.  Returns the table."""
    table = gradient_table(layout, grads, g_data, rg)
    for name, e in table.items():
        assert e["max_over_maxnorm"] < maxn, (where, name, "total, max-norm", e)
        assert e["rel_l2"] < rel, (where, name, "total, relative L2", e)
        if "data_rel_l2" in e:
            assert e["data_err_max"] <= maxn * e["data_ref_max"] + F32_ALLOWANCE * e["reg_body_max"] + 1e-30, \
                (where, name, "back-propagated part, max-norm", e)
            assert e["data_err_norm"] <= rel * e["data_ref_norm"] + F32_ALLOWANCE * e["reg_body_norm"] + 1e-30, \
                (where, name, "back-propagated part, relative L2", e)
            assert e["row0_err"] <= rel * e["row0_data_norm"] + F32_ALLOWANCE * e["row0_reg_norm"] + 1e-30, \
                (where, name, "back-propagated part, row 0", e)
    return table
