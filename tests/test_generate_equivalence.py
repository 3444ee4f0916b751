"""`Rater.generate` keeps its beam on float keys and drops continuations that cannot reach the kept 256
(rater.py); this must not change the result of the reference's procedure -- insort EVERY continuation by
cost, then truncate (rating.py:685-709) -- which is restated here naively and run side by side."""
from bisect import insort_left

import numpy as np
import pytest

from ocrd_keraslm_amd.lib import Node, Rater
from tests.oracle_engine import OracleLM


def naive_generate(r, prefix, length, context, variants):
    state = None
    for char in prefix[:-1]:
        _, states = r._predict_refs([char], [state], context)
        state = states[0]
    next_fringe = [Node(state=state, value=prefix[-1], cost=0.0)]
    i_c = r.mapping[1]
    for _ in range(length):
        fringe = next_fringe
        preds, states = r._predict_refs([n.value for n in fringe], [n.state for n in fringe], context)
        next_fringe = []
        for j, n in enumerate(fringe):
            pred = preds[j]
            pred_best = np.argsort(pred)[-10:]
            pred_best = pred_best[np.searchsorted(pred[pred_best], 0.004):]
            costs = -np.log(pred[pred_best])
            for best, cost in zip(pred_best, costs):
                if best not in i_c:
                    continue
                insort_left(next_fringe, Node(parent=n, state=states[j], value=i_c[best], cost=cost))
        next_fringe = next_fringe[:256]
    return [''.join([n.value for n in res.to_sequence()]) for res in next_fringe[0:variants]], \
        [float(n.cum_cost) for n in next_fringe]


@pytest.mark.parametrize("seed,emb_std", [(1, 0.05), (2, 0.5), (3, 1.5)])
def test_generate_equals_naive_beam(seed, emb_std):
    chars = [chr(c) for c in range(0x41, 0x41 + 60)]          # 60 types: the fringe overflows 256 within two steps
    r = Rater(engine_factory=OracleLM)
    r.width, r.depth, r.length = 16, 1, 8
    r.stateful, r.incremental = False, True
    r.mapping = ({c: i + 1 for i, c in enumerate(chars)}, {i + 1: c for i, c in enumerate(chars)})
    r.voc_size = len(chars) + 1
    r.configure()
    r.model.init_weights(seed=seed, emb_std=emb_std)
    r.status = 2
    want, costs = naive_generate(r, "AB", 7, [17], 12)
    assert len(costs) == 256                                   # the truncation really was exercised
    got = r.generate("AB", 7, [17], 12)
    assert got == want
