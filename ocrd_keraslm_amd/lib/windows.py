"""Host-side data preparation of the Rater: text -> index windows.

Behavioural restatement of the stateful branch of `Rater._gen_data` /
`Rater._vectorize` / `Rater._gen_data_from_files`
(ocrd_keraslm/lib/rating.py:977-1158), producing int32 index arrays for the HIP
engine instead of one-hot booleans:

  * windows advance by `length`; window k holds text[kT:(k+1)T] as input and the
    same span shifted by one as target (rating.py:1050-1052);
  * the residue after the last full window is emitted as ONE extra window holding
    text[i:size-1] -> text[i+1:size], right-padded with index 0 for characters AND
    contexts (rating.py:1096-1102, 1123-1143); padded target positions are all-zero
    one-hot rows there, encoded here as -1;
  * unmapped characters become index 0 and are reported through `on_unmapped`
    (rating.py:1133-1135, 1149-1151);
  * training augmentation (rating.py:1062-1077): after each full window, driven by
    one uniform number that is re-scaled and re-used: with probability
    `char_degradation` the window is repeated with one input column zeroed, with
    probability `context_degradation` it is repeated with one context zeroed;
  * the context of a file is ceil(year/10) parsed from `author_title_year.ext`
    names, else 0 (rating.py:993-999).

`stateless_batches` restates the non-stateful branch of the same functions (windows of
`length` characters predicting ONE next character, rating.py:1040-1102): see there.
"""
from __future__ import annotations

import os
import unicodedata
from math import ceil

import numpy as np

CTX_VOCAB = 200


def normalize(text):
    return unicodedata.normalize('NFC', text)


def read_normalize_file(file):
    """rating.py:1320-1323"""
    text = normalize(file.read())
    return text, len(text)


def context_from_filename(name):
    """rating.py:993-999 -- `a_b_1784.txt` -> [ceil(1784/10)], anything else -> [0]."""
    parts = os.path.basename(name).split('.')[0].split('_')
    if len(parts) == 3:
        return [ceil(int(parts[2]) / 10)]
    return [0]


def clamp_context(context):
    """ceil(year/10) reaches 200 for years >= 1991, outside Embedding(200,10)
    (latent bugs): clamp instead of reading out of bounds."""
    return [min(max(int(c), 0), CTX_VOCAB - 1) for c in context]


N_CODEPOINTS = 0x110000


def codepoints(text):
    """the code points of a string as one uint32 vector (lone surrogates pass through)"""
    if not text:
        return np.zeros(0, dtype=np.uint32)
    try:
        return np.frombuffer(text.encode('utf-32-le', 'surrogatepass'), dtype='<u4')
    except UnicodeEncodeError:       # pragma: no cover
        return np.array([ord(ch) for ch in text], dtype=np.uint32)


def full_lookup_table(c_i):
    """code point -> id over ALL code points (-1 = unmapped): a gather through it needs no bounds handling"""
    t = np.full(N_CODEPOINTS, -1, dtype=np.int32)
    for ch, k in c_i.items():
        if len(ch) == 1:
            t[ord(ch)] = k
    return t


def encode(text, c_i, on_unmapped=None, base=0):
    """characters -> int32 ids (0 = unmapped).  Vectorised: code points index a look-up table of the
    mapped characters; only the (rare) unmapped positions go through Python, to be reported."""
    n = len(text)
    if n == 0:
        return np.zeros(0, dtype=np.int32)
    if n < 64 or not c_i:            # short strings: the plain loop is cheaper than building the table
        ids = np.zeros(n, dtype=np.int32)
        for j, char in enumerate(text):
            k = c_i.get(char)
            if k is None:
                if on_unmapped is not None:
                    on_unmapped(char, base + j)
            else:
                ids[j] = k
        return ids
    cps = codepoints(text)
    lut = _lookup_table(c_i)
    if int(cps.max()) < len(lut):    # (the common case -- every code point inside the table: one gather)
        ids = lut[cps]
    else:
        inside = cps < len(lut)
        ids = np.where(inside, lut[np.where(inside, cps, 0)], -1).astype(np.int32)
    hit = ids >= 0
    if hit.all():
        return ids
    ids[~hit] = 0
    if on_unmapped is not None:
        for j in np.nonzero(~hit)[0]:
            on_unmapped(text[int(j)], base + int(j))
    return ids


_TABLES = {}


def _lookup_table(c_i):
    """code point -> id (-1 = unmapped) as one array; cached by content"""
    key = hash(frozenset(c_i.items()))
    t = _TABLES.get(key)
    if t is None:
        single = [(ord(ch), k) for ch, k in c_i.items() if len(ch) == 1]
        t = np.full((max(a for a, _ in single) + 1) if single else 1, -1, dtype=np.int32)
        for a, b in single:
            t[a] = b
        if len(_TABLES) > 64:
            _TABLES.clear()
        _TABLES[key] = t
    return t


def count_windows(size, length):
    """number of batches `stateful_windows` yields (== the reference's epoch sizes
    ceil((size-1)/length) of rating.py:487, 515 for size >= 2)."""
    full = len(range(length, size, length))
    last = (length + (full - 1) * length) if full else 0
    return full + (1 if last + 1 < size else 0)


def stateful_windows(text, context, length, c_i, train=False, rng=None, char_degradation=0.01,
                     context_degradation=0.1, on_unmapped=None):
    """Yield (x[T], ctx[T,C], y[T]) int32 arrays for one text, in order."""
    size = len(text)
    context = clamp_context(context)
    n_ctx = len(context)
    ids = encode(text, c_i, on_unmapped)
    i = 0
    for i in range(length, size, length):
        x = ids[i - length:i].copy()
        y = ids[i - length + 1:i + 1].copy()
        z = np.tile(np.asarray(context, dtype=np.int32), (length, 1)) if n_ctx else np.zeros((length, 0), np.int32)
        yield x, z, y
        if train:
            rand = float(rng.uniform(0, 1)) if rng is not None else float(np.random.uniform(0, 1, 1)[0])
            rand_max = char_degradation
            if 0 < rand < rand_max:
                j = int((length - 1) * rand / rand_max)
                xa = x.copy()
                xa[j] = 0
                yield xa, z, y
            rand = (rand - rand_max) / (1 - rand_max)
            rand_max = context_degradation
            if 0 < rand < rand_max and n_ctx:
                j = int(n_ctx * rand / rand_max)      # == int((len(x)-1)*rand/rand_max)+1 over [chars]+contexts
                za = z.copy()
                za[:, min(j, n_ctx - 1)] = 0
                yield x, za, y
    if i + 1 < size:
        n = size - 1 - i
        x = np.zeros(length, dtype=np.int32)
        y = np.full(length, -1, dtype=np.int32)
        z = np.zeros((length, n_ctx), dtype=np.int32)
        x[:n] = ids[i:size - 1]
        y[:n] = ids[i + 1:size]
        if n_ctx:
            z[:n] = np.asarray(context, dtype=np.int32)
        yield x, z, y


def file_windows(files, length, c_i, train=False, repeat=False, rng=None, on_new_file=None, on_unmapped=None,
                 char_degradation=0.01, context_degradation=0.1):
    """rating.py:977-1002: windows of a list of open text files; `on_new_file(name)`
    is the reset hook of stateful training (callbacks.py:50-60)."""
    while True:
        for file in files:
            file.seek(0)
            if on_new_file is not None:
                on_new_file(file.name)
            text, _ = read_normalize_file(file)
            yield from stateful_windows(text, context_from_filename(file.name), length, c_i, train=train, rng=rng,
                                        char_degradation=char_degradation, context_degradation=context_degradation,
                                        on_unmapped=on_unmapped)
        if not repeat:
            break


def _vectorize_stateless(sequences, next_ids, context, length, batch_size, c_i, on_unmapped):
    """rating.py:1104-1158 for stateful=False: sequences RIGHT-padded with id 0 to `length`, contexts only under
    the characters, one target id per row (0 = unmapped, -1 = row without a target)"""
    n_ctx = len(context)
    x = np.zeros((batch_size, length), dtype=np.int32)
    z = np.zeros((batch_size, length, n_ctx), dtype=np.int32)
    y = np.full(batch_size, -1, dtype=np.int32)
    for i, seq in enumerate(sequences):
        assert i < batch_size and len(seq) <= length
        ids = encode(seq, c_i, on_unmapped, base=i * length)
        x[i, :len(ids)] = ids
        if n_ctx:
            z[i, :len(ids)] = np.asarray(context, dtype=np.int32)
        if next_ids is not None and i < len(next_ids) and len(seq) > 0:
            y[i] = next_ids[i]      # (the reference sets the one-hot target inside its per-character loop,
            #                          so a row with an EMPTY input sequence has no target at all)
    return x, z, y


def stateless_batches(text, context, length, c_i, steps, batch_size=128, train=False, split=None, validation_split=0.2,
                      variable_length=True, first_window=0.1, char_degradation=0.01, context_degradation=0.1,
                      on_unmapped=None):
    """Stateless windows of one text (rating.py:1004-1102 with stateful=False): window i holds
    text[i-length:i] and predicts text[i], i = 0, steps, 2*steps, ...; yields (x [b,L'], ctx [b,L',C], y [b]).

      * windows with i < length are partial: in prediction mode each is its own batch of 1 (of length i
        when `variable_length`, else right-padded to `length`); in training mode they join the batches;
      * with `split` (one uniform number per window position) the training generator keeps the
        positions whose number is >= validation_split and the validation generator the others;
      * training augmentation after each FULL batch, driven by the last window's random number,
        re-scaled and re-used: one character column zeroed for the whole batch (`char_degradation`),
        the context zeroed (`context_degradation`), and with rate `first_window` a shortened
        (variable length: last j columns) or left-erased copy of the batch;
      * the characters behind the last window position form one more single-row batch.
    Random numbers come from numpy's global generator, as in the reference."""
    size = len(text)
    context = clamp_context(context)
    n_ctx = len(context)

    def target(ch):
        k = c_i.get(ch)
        if k is None:
            if on_unmapped is not None:
                on_unmapped(ch, -1)
            return 0
        return k
    sequences, nexts = [], []
    i = 0
    for i in range(0, size, steps):
        if isinstance(split, np.ndarray):
            if (split[int(i / steps)] < validation_split) == train:
                continue
            rand = (split[int(i / steps)] - validation_split) / (1 - validation_split)
        else:
            rand = np.random.uniform(0, 1, 1)[0]
        if i < length:
            if train:
                sequences.append(text[0:i])
            else:
                yield _vectorize_stateless([text[0:i]], [target(text[i])], context, (i if variable_length else length) or length,
                                           1, c_i, on_unmapped)
                continue
        else:
            sequences.append(text[i - length:i])
        nexts.append(target(text[i]))
        if len(sequences) % batch_size == 0:
            x, z, y = _vectorize_stateless(sequences, nexts, context, length, batch_size, c_i, on_unmapped)
            yield x, z, y
            sequences, nexts = [], []
            if train:
                rand_max = char_degradation
                if 0 < rand < rand_max:
                    j = int((length - 1) * rand / rand_max)
                    xa = x.copy()
                    xa[:, j] = 0
                    yield xa, z, y
                rand = (rand - rand_max) / (1 - rand_max)
                rand_max = context_degradation
                if 0 < rand < rand_max:
                    j = int(n_ctx * rand / rand_max)          # == int((len(x)-1)*rand/rand_max)+1 over [chars]+contexts
                    za = z.copy()
                    if n_ctx:
                        za[:, :, min(j, n_ctx - 1)] = 0
                    yield x, za, y
                rand = (rand - rand_max) / (1 - rand_max)
                rand_max = first_window
                if 0 < rand < rand_max:
                    j = int((length - 1) * rand / rand_max) + 1
                    if variable_length:
                        yield x[:, -j:].copy(), z[:, -j:].copy(), y
                    else:
                        xa = x.copy()
                        xa[:, 0:j] = 0
                        yield xa, z, y
    if sequences:
        yield _vectorize_stateless(sequences, nexts, context, length, len(sequences), c_i, on_unmapped)
    if i + 1 < size:
        yield _vectorize_stateless([text[i:size - 1]], [target(text[size - 1])], context, length, 1, c_i, on_unmapped)


def stateless_file_batches(files, length, c_i, steps, repeat=False, **kwargs):
    """rating.py:977-1002 for stateful=False: the batches of a list of open text files"""
    while True:
        for file in files:
            file.seek(0)
            text, _ = read_normalize_file(file)
            yield from stateless_batches(text, context_from_filename(file.name), length, c_i, steps, **kwargs)
        if not repeat:
            break
