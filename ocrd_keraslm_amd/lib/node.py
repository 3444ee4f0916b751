"""Beam-search tree node (behavioural mirror of ocrd_keraslm/lib/rating.py:1240-1318).

Observable contract reproduced here:
  * `cum_cost` accumulates from the parent; `length` counts nodes from the root;
  * ordering AND equality compare `pro_cost()` = cum_cost + 0.5 * (characters of
    `extras[1].Unicode` not yet consumed by `value`), so `node in list` and
    `list.remove(node)` match any node of equal prospective cost;
  * `to_sequence(stop_at)` walks to the root and returns root..self, or -- with
    `stop_at` -- root..(nearest ancestor contained in stop_at); the first
    non-empty result is cached until `cut_at` clears it;
  * `cut_at(node)` detaches the chain just above `node`.

`state` is opaque to callers (rate.py:265, 290 hand it back untouched).  On the
HIP path it is a `StateRef` (a reference-counted slot of the device state pool,
see rater.py); the reference's list-of-arrays form is still accepted by
`Rater.predict`.
"""


class Node(object):
    __hash__ = None   # comparison by cost makes nodes unhashable, as in the reference

    def __init__(self, state, value, cost, parent=None, extras=None):
        self.value = value
        self.parent = parent
        self.state = state
        self.cum_cost = cost if parent is None else parent.cum_cost + cost
        self.length = 1 if parent is None else parent.length + 1
        self.extras = extras
        self._sequence = None

    def to_sequence(self, stop_at=None):
        if not self._sequence:
            chain = []
            collecting = not stop_at
            cursor = self
            while cursor:
                if stop_at and cursor in stop_at:
                    collecting = True
                if collecting:
                    chain.append(cursor)
                cursor = cursor.parent
            chain.reverse()
            self._sequence = chain
        return self._sequence

    def cut_at(self, node):
        cursor = self
        while cursor:
            if cursor.parent is node:
                cursor.parent = None
                self._sequence = None
                return
            cursor = cursor.parent

    def pro_cost(self):
        pending = len(self.extras[1].Unicode) - len(self.value) if self.extras else 0
        return self.cum_cost + 0.5 * pending

    def __lt__(self, other):
        return self.pro_cost() < other.pro_cost()

    def __le__(self, other):
        return self.pro_cost() <= other.pro_cost()

    def __eq__(self, other):
        return self.pro_cost() == other.pro_cost()

    def __ne__(self, other):
        return self.pro_cost() != other.pro_cost()

    def __gt__(self, other):
        return self.pro_cost() > other.pro_cost()

    def __ge__(self, other):
        return self.pro_cost() >= other.pro_cost()
