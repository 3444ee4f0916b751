"""`streams.StreamBatcher` (B streams advanced together, what `Rater.train` uses) against one `windows.file_windows`
generator per stream (the restatement of rating.py:977-1102 that tests/test_stateful_train_golden.py pins to fixtures of
the reference): the same batches -- window contents, zero-padded tails, the train=True augmented copies under the same
random generator, the streams that enter a new file -- for several hundred steps, with files of very different lengths
(shorter than a window, exactly a multiple of it, no tail, tail of one character)."""
import io

import numpy as np
import pytest

from ocrd_keraslm_amd.lib import streams, windows


class MemFile(object):
    def __init__(self, name, text):
        self.name = name
        self.text = text
        self._f = io.StringIO(text)

    def read(self):
        return self._f.read()

    def seek(self, pos):
        return self._f.seek(pos)


def make_files(rng, n, T, with_year=True):
    chars = "abcdefghij klmnop\nqrs"
    out = []
    sizes = [1, 2, T - 1, T, T + 1, T + 2, 2 * T, 2 * T + 1, 3 * T + 5, 7 * T - 3]
    for k in range(n):
        size = sizes[k % len(sizes)] if k < 2 * len(sizes) else int(rng.integers(2, 6 * T))
        text = "".join(chars[j] for j in rng.integers(0, len(chars), size))
        name = ("a_b%d_%d.txt" % (k, 1700 + 7 * k)) if with_year else ("plain%d.txt" % k)
        out.append(MemFile(name, text))
    return out


@pytest.mark.parametrize("train,char_deg,ctx_deg,with_year", [(False, 0.01, 0.1, True), (True, 0.3, 0.4, True), (True, 0.01, 0.1, True),
                                                            (True, 0.5, 0.0, False), (True, 0.0, 0.9, True)])
def test_batcher_equals_one_generator_per_stream(train, char_deg, ctx_deg, with_year):
    T, B = 8, 7
    frng = np.random.default_rng(5)
    files = make_files(frng, 37, T, with_year)
    c_i = {c: i + 1 for i, c in enumerate(sorted(set("abcdefghij klmno\nqrs")))}      # ('p' is unmapped)
    per_stream = [files[s::B] for s in range(B)]
    # the reference: one generator per stream, pulled stream by stream
    rng_a = np.random.default_rng(11)
    resets_a = []
    gens = []
    for s in range(B):
        gens.append(windows.file_windows(per_stream[s], T, c_i, train=train, repeat=True, rng=rng_a,
                                         on_new_file=(lambda name, s=s: resets_a.append(s)),
                                         char_degradation=char_deg, context_degradation=ctx_deg))
    rng_b = np.random.default_rng(11)
    bat = streams.StreamBatcher(per_stream, T, c_i, train=train, rng=rng_b, char_degradation=char_deg, context_degradation=ctx_deg)
    for step in range(300):
        del resets_a[:]
        xs, zs, ys = [], [], []
        for g in gens:
            x, z, y = next(g)
            xs.append(x); zs.append(z); ys.append(y)
        (x, z, y), rows = bat.next_batch()
        assert np.array_equal(x, np.stack(xs)), step
        assert np.array_equal(y, np.stack(ys)), step
        assert np.array_equal(z, np.stack(zs)), step
        assert sorted(rows) == sorted(resets_a), step
    # both paths consumed the random stream identically (a generator draws the number of a window when it is resumed, the
    # batcher right behind the window: one more pull brings the generators level)
    for g in gens:
        next(g)
    assert rng_a.uniform() == rng_b.uniform()


def test_batcher_on_the_golden_file_set():
    """the files of the reference-generated fixture (tests/golden/stateful_train.json), all in one stream each"""
    import json
    import os
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stateful_train.json")))
    files = [MemFile(f["name"], f["text"]) for f in G["files"]]
    chars = sorted(set("".join(f["text"] for f in G["files"])))
    c_i = {c: i + 1 for i, c in enumerate(chars)}
    T = G["length"]
    per_stream = [[f] for f in files]
    gens = [windows.file_windows(fs, T, c_i, train=False, repeat=True) for fs in per_stream]
    bat = streams.StreamBatcher(per_stream, T, c_i, train=False)
    for step in range(40):
        want = [next(g) for g in gens]
        (x, z, y), _rows = bat.next_batch()
        assert np.array_equal(x, np.stack([w[0] for w in want]))
        assert np.array_equal(z, np.stack([w[1] for w in want]))
        assert np.array_equal(y, np.stack([w[2] for w in want]))


def test_batcher_reports_unmapped_characters_once_per_occurrence():
    """the batcher maps all files in one pass: every character outside the mapping is reported with its position in its
    file (windows.encode reports the same positions when a generator encodes the file)"""
    T = 8
    files = [MemFile("a_b_1800.txt", "abcpabcxab" * 3), MemFile("a_c_1810.txt", "ab" * 9), MemFile("a_d_1820.txt", "pppa" * 5)]
    c_i = {"a": 1, "b": 2, "c": 3}
    want = []
    for f in files:
        windows.encode(f.text, c_i, on_unmapped=lambda ch, pos, f=f: want.append((f.name, ch, pos)))
    got = []
    bat = streams.StreamBatcher([[f] for f in files], T, c_i,
                                codepoints={id(f): windows.codepoints(f.text) for f in files[:2]},
                                on_unmapped=lambda ch, pos: got.append((ch, pos)))
    bat.prepare()
    assert got == [(ch, pos) for _n, ch, pos in want] and len(got) == 6 + 15
    (x, _z, _y), _rows = bat.next_batch()
    assert x[0].tolist() == [1, 2, 3, 0, 1, 2, 3, 0] and x[2].tolist() == [0, 0, 0, 1, 0, 0, 0, 1]
