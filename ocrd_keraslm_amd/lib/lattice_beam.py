"""Beam bookkeeping of the lattice decoder `Rater.rate_best` (behaviour of rating.py:712-885), organised around
one table per lattice edge instead of a tree node per hypothesis and character.

What the reference does on an edge: every (incoming hypothesis, alternative) pair becomes a node that is
re-inserted into a cost-sorted list after every character it consumes; the list is worked off from its worst end
in batches of `batch_size`, finished nodes move to the destination's beam, and two cost margins prune
(+2.5 against the best waiting node, +15 against the best finished one).

Here such a pair is a TRACK: row i of a few parallel arrays (parent hypothesis, alternative, characters
consumed, accumulated cost, state handle).  A track never changes identity while it walks through its
alternative's text, so

  * the per-character work of a batch is array arithmetic: target ids and confidence terms are looked up from
    per-alternative tables built once per edge, the language-model costs of a whole batch come from one fancy
    index into the probability matrix, costs / positions / sort keys are updated in place;
  * the two ordered lists hold track numbers and float keys (bisect on the keys; `insort_left` = ties go in
    front, which is what decides between equally expensive hypotheses and therefore pinned by fixtures);
  * a batch is cut off the worst end of the waiting list as a slice: nodes popped one by one from the end and
    insorted into an empty list end up in their original relative order (a stable sort of the slice);
  * `Node` objects -- what callers see in tracebacks -- are built only for the `beam_width` tracks that survive at
    the destination.

The order-sensitive parts stay scalar loops on purpose: whether a track is re-queued depends on the head of the
waiting list as left by the tracks re-queued before it, and history clustering removes beam entries while it runs.
"""
from __future__ import annotations

from bisect import bisect_left
from math import log

import numpy as np

from .node import Node

WAITING_MARGIN = 2.5      # rating.py:847
FINISHED_MARGIN = 15.0    # rating.py:816
LOOKAHEAD = 0.5           # Node.pro_cost: half a bit per character still to come (rating.py:1300)


class FinishedBeam(object):
    """The destination node's beam while an edge is decoded: entries ordered by cost, `insort_left` ties.
    An entry is either a Node that another in-edge left there or a finished track of this edge."""

    def __init__(self, nodes):
        self.keys = [n.pro_cost() for n in nodes]
        self.items = [("node", n) for n in nodes]

    def __len__(self):
        return len(self.keys)

    def best_cost(self, tracks):
        kind, ref = self.items[0]
        return ref.cum_cost if kind == "node" else tracks.cum[ref]

    def insert(self, key, item):
        pos = bisect_left(self.keys, key)
        self.keys.insert(pos, key)
        self.items.insert(pos, item)

    def remove_first_with_key(self, key):
        """`list.remove(node)` with cost-equality (rating.py:1308-1312): the FIRST entry of equal cost goes"""
        pos = self.keys.index(key)
        del self.keys[pos]
        del self.items[pos]

    def value_state_cost(self, i, tracks):
        kind, ref = self.items[i]
        if kind == "node":
            return ref.value, ref.state, ref.cum_cost
        return tracks.text[tracks.alt[ref]], tracks.state[ref], tracks.cum[ref]


class EdgeTracks(object):
    """All (incoming hypothesis x alternative) pairs of one lattice edge.

    Plain Python lists, not arrays: an edge has a few dozen tracks (beam width x alternatives), and at that size every numpy
    call costs more than the arithmetic it does -- the bookkeeping of an edge was 70 of the 110 microseconds a lattice edge took
    with the model's step at 40 (round 4).  All costs are Python floats = IEEE doubles, the arithmetic and its order are those
    of the array form (and of the reference): ties and the two margins are decided on identical values."""

    def __init__(self, incoming, alternatives, element, c_i, lm_weight, logger):
        n_in, n_alt = len(incoming), len(alternatives)
        self.incoming = incoming
        self.alternatives = alternatives
        self.element = element
        self.text = [a.Unicode for a in alternatives]
        self.length = [len(t) for t in self.text]
        self.lm_weight = lm_weight
        # per alternative: ids of its characters (0 = unmapped; reported when a track first reaches one, once per
        # alternative and character, rating.py:830-837) and the confidence term every one of its characters costs
        # (rating.py:839-840)
        get = c_i.get
        self.ids = [[get(char, 0) for char in t] for t in self.text]
        self.unmapped = [[char not in c_i for char in t] for t in self.text]
        self.reported = [set() for _ in alternatives]
        self.logger = logger
        self.conf_term = [-log(max(a.conf, 1e-99), 2) * (1. - lm_weight) for a in alternatives]
        # the tracks, in the order the reference creates its nodes: hypothesis-major, alternative-minor
        self.parent = [p for p in range(n_in) for _ in range(n_alt)]
        self.alt = list(range(n_alt)) * n_in
        self.pos = [0] * (n_in * n_alt)
        self.cum = [float(h.cum_cost) for h in incoming for _ in range(n_alt)]
        self.state = [h.state for h in incoming for _ in range(n_alt)]

    def __len__(self):
        return len(self.parent)

    def keys(self, rows):
        """prospective cost: what the lists are ordered by"""
        cum, length, alt, pos = self.cum, self.length, self.alt, self.pos
        return [cum[i] + LOOKAHEAD * (length[alt[i]] - pos[i]) for i in rows]

    def last_chars(self, rows):
        """the character each track feeds into the model next: its own last one, or -- nothing consumed yet -- the
        last character of the hypothesis it continues (rating.py:822)"""
        out = []
        for i in rows:
            p = self.pos[i]
            out.append(self.text[self.alt[i]][p - 1] if p else self.incoming[self.parent[i]].value[-1])
        return out

    def targets(self, rows):
        """id of the character each track of the batch consumes next (the one whose probability `advance` looks at)"""
        ids, alt, pos = self.ids, self.alt, self.pos
        return [ids[alt[i]][pos[i]] for i in rows]

    def advance(self, rows, probs, new_states, target=None):
        """consume one character on every track of the batch; probs: [n, V] rows, or [n] -- already the probabilities
        of `targets(rows)` (`target`: that list, if the caller has it)"""
        if target is None:
            target = self.targets(rows)
        alt, pos, cum, state = self.alt, self.pos, self.cum, self.state
        if not all(target):
            for k, i in enumerate(rows):
                a, p = alt[i], pos[i]
                if self.unmapped[a][p] and self.text[a][p] not in self.reported[a]:
                    self.reported[a].add(self.text[a][p])
                    self.logger.error('unmapped character "%s" at input alternative %d of element %s', self.text[a][p],
                                      self.alternatives[a].index or k, self.element.id if self.element else "space")
        probs = np.asarray(probs, dtype=np.float64)
        p_next = (probs if probs.ndim == 1 else probs[np.arange(len(rows)), np.asarray(target)]).tolist()
        w, conf = self.lm_weight, self.conf_term
        # (math.log(p, 2) element by element, as the reference computes it (rating.py:843): np.log2 -- and numpy's own log --
        #  can differ from libm in the last bit, and ties and the +2.5 / +15 margins are decided on exact values)
        for i, p, s in zip(rows, p_next, new_states):
            cum[i] += -log(p if p > 1e-99 else 1e-99, 2) * w + conf[alt[i]]
            pos[i] += 1
            state[i] = s

    def node(self, i):
        """the tree node of a finished track (only the survivors get one)"""
        alt = self.alternatives[self.alt[i]]
        parent = self.incoming[self.parent[i]]
        n = Node(parent=parent, state=self.state[i], value=alt.Unicode, cost=0.0, extras=(self.element, alt))
        n.cum_cost = self.cum[i]      # the running sum itself, not parent + difference (last-bit identical)
        return n


def decode_edge(tracks, finished, predict, batch_size, max_batches, close_states=None):
    """Walk all tracks of an edge through their alternatives (rating.py:796-851).

    predict(last_chars, states, targets) -> (probs, new states) with probs [n, V] or -- an engine that delivers only what
    is looked at -- [n], the probabilities of the characters `targets`; close_states(a, b) -> whether two state handles are
    within the clustering distance (None: no history clustering).  Finished tracks end up in `finished`."""
    # the waiting list: track numbers + keys.  It starts in creation order (UNSORTED, as in the reference, whose first
    # batch is therefore cut off the end of the creation order) and is kept sorted from the first re-queueing on.
    waiting = list(range(len(tracks)))
    wkeys = tracks.keys(waiting)
    pos, length, alt, cum, state = tracks.pos, tracks.length, tracks.alt, tracks.cum, tracks.state
    for _ in range(max_batches):
        # ---- cut the next batch off the worst end; finished tracks found on the way move to the destination beam
        batch, taken = [], 0
        for i in reversed(waiting):
            taken += 1
            if pos[i] == length[alt[i]]:
                _finish(tracks, finished, i, close_states)
            else:
                batch.append(i)
                if len(batch) >= batch_size:
                    break
        del waiting[len(waiting) - taken:]
        del wkeys[len(wkeys) - taken:]
        if not batch:
            break
        batch.reverse()                                           # back to list order, then ordered by key:
        bkeys = tracks.keys(batch)
        batch = [batch[k] for k in sorted(range(len(batch)), key=bkeys.__getitem__)]     # insort_left of nodes popped from the end = a stable sort
        if len(finished) and cum[batch[0]] >= finished.best_cost(tracks) + FINISHED_MARGIN:
            break
        # ---- one character on every track of the batch
        target = tracks.targets(batch)
        probs, new_states = predict(tracks.last_chars(batch), [state[i] for i in batch], target)
        tracks.advance(batch, probs, new_states, target)
        # ---- back into the waiting list, unless hopeless against its current head
        for i, key in zip(batch, tracks.keys(batch)):
            if waiting and cum[i] >= cum[waiting[0]] + WAITING_MARGIN:
                state[i] = None
                continue
            at = bisect_left(wkeys, key)
            wkeys.insert(at, key)
            waiting.insert(at, i)
        del waiting[max_batches * batch_size:]
        del wkeys[max_batches * batch_size:]


def _finish(tracks, finished, i, close_states):
    """a track has consumed its alternative: history clustering against the destination beam, then insertion"""
    if close_states is not None:
        value, state, cost = tracks.text[tracks.alt[i]], tracks.state[i], tracks.cum[i]
        for k in range(len(finished)):
            other_value, other_state, other_cost = finished.value_state_cost(k, tracks)
            if value == other_value and close_states(state, other_state):
                if other_cost < cost:
                    return                                     # redundant: the cheaper twin is already there
                finished.remove_first_with_key(finished.keys[k])
                break
    finished.insert(tracks.cum[i], ("track", i))


def lattice_edges(graph, start):
    """edges in the order the decoder must visit them: destinations in topological order, sources already reached
    (rating.py:763-773)"""
    import networkx as nx
    reached = {start}
    pred = graph.pred
    for node in nx.topological_sort(graph):
        for source in pred[node]:      # (= graph.in_edges([node]) without a view object per node)
            if source in reached:
                yield source, node
                reached.add(node)


def advance_traceback(beam, traceback):
    """What `Rater.next_path` returns (rating.py:862-885): commit to the part of the best hypothesis that reaches back
    to the previous beam, report it, and re-root the hypotheses that share it.

    beam: hypotheses at the current end of the text, best first; traceback = (previous beam, previous start node).
    Returns (path [(element, alternative, score)], entropy of that path in bits, (beam re-rooted, new start node))."""
    previous_beam, previous_start = traceback
    committed = beam[0].to_sequence(stop_at=previous_beam)      # root .. the best hypothesis' ancestor in the previous beam
    anchor = committed[-1]

    def segment(node):
        element, alternative = node.extras
        before = node.parent.cum_cost if node.parent else previous_start.cum_cost
        return element, alternative, 2.0 ** (-(node.cum_cost - before) / len(alternative.Unicode))

    path = [segment(node) for node in committed if node.extras]
    # Hypotheses that do not reach the anchor are gone; the others are cut loose just above it and re-queued by cost
    # (equal costs: the later one in front).  The order of these two steps per hypothesis is part of the behaviour:
    # cutting one hypothesis detaches the anchor's child it passes through, so a later hypothesis that shares that
    # child no longer reaches the anchor and is dropped as well.
    keys, rerooted = [], []
    for hyp in beam:
        if not hyp.to_sequence(stop_at=[anchor]):
            continue
        hyp.cut_at(anchor)
        at = bisect_left(keys, hyp.pro_cost())
        keys.insert(at, hyp.pro_cost())
        rerooted.insert(at, hyp)
    return path, anchor.cum_cost - previous_start.cum_cost, (rerooted, anchor)
