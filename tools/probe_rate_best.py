"""Diagnostic: wall time (and, from the command line, the host profile) of Rater.rate_best on a synthetic page lattice
(N glyph edges, up to 3 alternatives each, white-space pseudo edges every few glyphs).
  python tools/probe_rate_best.py [N]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')

CHARS = "abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ.,;\n-"


def make_rater(width=512, depth=2):
    from ocrd_keraslm_amd.lib import Rater
    r = Rater()
    r.width, r.depth, r.length = width, depth, 256
    r.stateful, r.incremental = False, True
    r.mapping = ({c: i + 1 for i, c in enumerate(CHARS)}, {i + 1: c for i, c in enumerate(CHARS)})
    r.voc_size = len(CHARS) + 1
    r.configure()
    r.model.init_weights(seed=3, emb_std=0.5)
    r.status = 2
    return r


def page(rng, N):
    import networkx as nx
    from ocrd_keraslm_amd.wrapper.lattice import PlainTextEquiv as TE
    g = nx.DiGraph(level='glyph')
    g.add_node(0)
    for k in range(N):
        g.add_node(k + 1)
        if k % 6 == 5:
            alts, elem = [TE(' ', 1.0)], None
        else:
            n_alt = int(rng.integers(1, 4))
            cs = rng.choice(list(CHARS[:52]), size=n_alt, replace=False)
            conf = np.sort(rng.uniform(0.85, 0.95, n_alt))[::-1]
            alts, elem = [TE(str(c), float(p)) for c, p in zip(cs, conf)], object()
        g.add_edge(k, k + 1, element=elem, alternatives=alts)
    return g


def run(N=600, pages=3, clustering=0, profile=False, verbose=False):
    """-> ms per lattice edge of the last page (the first pages warm up kernels, pools and buffers)"""
    r = make_rater()
    rng = np.random.default_rng(0)
    tb, pr, ms = None, None, None
    for k in range(pages):
        g = page(rng, N)
        if profile and k == pages - 1:
            import cProfile
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.time()
        path, entropy, tb = r.rate_best(g, 0, N, start_traceback=tb, context=[179], lm_weight=0.5, beam_width=10,
                                        beam_clustering_dist=clustering)
        dt = time.time() - t0
        if pr is not None:
            pr.disable()
        ms = dt / N * 1e3
        if verbose:
            print(f"page {k}: {N} edges in {dt:.2f} s = {ms:.3f} ms/edge, decided path {len(path)} entropy {entropy:.1f}")
    if pr is not None:
        import pstats
        pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
    return ms


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    for clustering in (0, 5):
        print("beam_clustering_dist =", clustering)
        run(N, clustering=clustering, profile=True, verbose=True)
