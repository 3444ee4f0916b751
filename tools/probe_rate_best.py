"""Diagnostic: wall time and host profile of Rater.rate_best on a synthetic page lattice
(N glyph edges, up to 3 alternatives each, white-space pseudo edges every few glyphs)."""
import cProfile
import pstats
import sys
import time

import networkx as nx
import numpy as np

sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import Rater
from ocrd_keraslm_amd.wrapper.lattice import PlainTextEquiv as TE

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
chars = "abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ.,;\n-"
r = Rater()
r.width, r.depth, r.length = 512, 2, 256
r.stateful, r.incremental = False, True
r.mapping = ({c: i + 1 for i, c in enumerate(chars)}, {i + 1: c for i, c in enumerate(chars)})
r.voc_size = len(chars) + 1
r.configure()
r.model.init_weights(seed=3, emb_std=0.5)
r.status = 2
rng = np.random.default_rng(0)


def page():
    g = nx.DiGraph(level='glyph')
    g.add_node(0)
    for k in range(N):
        g.add_node(k + 1)
        if k % 6 == 5:
            alts, elem = [TE(' ', 1.0)], None
        else:
            n_alt = int(rng.integers(1, 4))
            cs = rng.choice(list(chars[:52]), size=n_alt, replace=False)
            conf = np.sort(rng.uniform(0.85, 0.95, n_alt))[::-1]
            alts, elem = [TE(str(c), float(p)) for c, p in zip(cs, conf)], object()
        g.add_edge(k, k + 1, element=elem, alternatives=alts)
    return g


tb = None
for k in range(3):
    g = page()
    t0 = time.time()
    if k == 2:
        pr = cProfile.Profile()
        pr.enable()
    path, entropy, tb = r.rate_best(g, 0, N, start_traceback=tb, context=[179], lm_weight=0.5, beam_width=10,
                                    beam_clustering_dist=5)
    if k == 2:
        pr.disable()
    dt = time.time() - t0
    print(f"page {k}: {N} edges in {dt:.2f} s = {dt / N * 1e3:.2f} ms/edge, decided path {len(path)} entropy {entropy:.1f}")
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
