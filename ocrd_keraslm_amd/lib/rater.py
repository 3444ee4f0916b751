"""Rater: character-level LSTM language model for rating text -- MI355X edition.

Drop-in for `ocrd_keraslm.lib.Rater` (ocrd_keraslm/lib/rating.py:12-1238): same
attributes, same methods, same status machine and assertion conventions.  The
seam the reference fills with a compiled Keras model (`self.model`,
rating.py:56) is filled here by `engine.HipLM`, which runs the network on
hand-written gfx950 kernels; everything in this file is host logic
(vocabulary, windowing, beam bookkeeping, training loop control).

Differences that are deliberate:
  * `streams` (new, default 1): number of independent stateful streams trained
    in lockstep per GPU.  1 reproduces the reference's batch_size=1 stateful
    training (rating.py:90-92); larger values are the data-parallel capability
    north_star asks for (each stream has its own carried state and reset points).
  * beam searches keep hypothesis states in a device-resident pool; `Node.state`
    is then a `StateRef`.  `predict()` called directly still takes and returns
    the reference's list-of-arrays states (rating.py:622-639).
  * context ids are clamped to the embedding range (the reference can index
    Embedding(200,10) out of range for years >= 1991, rating.py:111, 996).
  * the stateless, non-incremental window mode (rating.py:93-99, 352-378) runs through the
    same engine with `set_window_mode(True)`: zero state per window, one target per window.
"""
from __future__ import annotations

import json
import logging
import signal
import time
from bisect import bisect_left, insort_left
from math import ceil, exp, log
from random import shuffle

import numpy as np

from . import lattice_beam, modelio, streams, windows
from .node import Node

PREC_BF16 = 1
PREC_SPLIT = 3


def _np(x):
    """engine results are torch tensors (HIP) or numpy arrays (test doubles)"""
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


class StateRef(object):
    """Handle of one hypothesis state in the engine's device pool.  Quacks like
    the reference's state list [h1,c1,...,hL,cL] of (1,W) arrays when indexed."""
    __slots__ = ("pool", "slot", "head", "__weakref__")

    def __init__(self, pool, slot):
        self.pool = pool
        self.slot = slot
        self.head = None      # host copy of the first `depth` state vectors (what history clustering compares)

    def __del__(self):
        try:
            self.pool.release(self.slot)
        except Exception:
            pass

    def __len__(self):
        return 2 * self.pool.depth

    def __getitem__(self, k):
        return self.pool.fetch(self.slot)[k][None, :]

    def __bool__(self):
        return True


class StatePool(object):
    """Slot allocator over engine.pool ([slots][2L][W] f32 in HBM)."""

    def __init__(self, engine, depth, initial=1024):
        self.engine = engine
        self.depth = depth
        self.capacity = 0
        self.free = []
        self._grow(initial)
        self.zero_slot = self.take()      # never released: the all-zero state of `None`
        self.engine.pool_zero([self.zero_slot])

    def _grow(self, n):
        new = max(n, 2 * self.capacity)
        self.engine.ensure_pool(new)
        self.free.extend(range(new - 1, self.capacity - 1, -1))
        self.capacity = new

    def take(self):
        if not self.free:
            self._grow(2 * self.capacity)
        return self.free.pop()

    def release(self, slot):
        self.free.append(slot)

    def ref(self):
        return StateRef(self, self.take())

    def refs(self, n):
        """n fresh handles at once (a beam search takes one per hypothesis and character)"""
        while len(self.free) < n:
            self._grow(2 * self.capacity)
        slots = self.free[-n:]
        del self.free[-n:]
        slots.reverse()                      # (the order `take` would have handed them out in)
        return [StateRef(self, slot) for slot in slots], slots

    def fetch(self, slot):
        return self.engine.pool_read([slot])[0]


class Rater(object):
    '''A character-level RNN language model for rating text (see module docstring).

    Interfaces (as the reference, rating.py:25-32):
    - `Rater.train` : file handles of character sequences
    - `Rater.test` : file handles of character sequences
    - `Rater.rate2` (alias `rate_once`) : character string, one by one
    - `Rater.rate` : character string, all at once
    - `Rater.rate_best` : lattice graph
    - `Rater.generate` : alternative list of characters and states
    '''

    def __init__(self, logger=None, engine_factory=None):
        # configuration variables (rating.py:39-47)
        self.width = 0
        self.depth = 0
        self.length = 0
        self.variable_length = True
        self.first_window = 0.1
        self.char_degradation = 0.01
        self.context_degradation = 0.1
        self.stateful = True
        self.mapping = ({}, {})
        # configuration constants (rating.py:49-51)
        self.batch_size = 128
        self.validation_split = 0.2
        self.smoothing = 0.2
        # runtime variables (rating.py:53-59)
        self.logger = logger or logging.getLogger(__name__)
        self.reset_cb = None
        self.incremental = False
        self.model = None
        self.history = {}
        self.status = 0
        self.voc_size = 0
        # MI355X additions
        self.streams = 1
        self.n_ctx = 1
        self.max_epochs = 100
        self.patience = 3
        self.seed = None
        self.device_dropout_masks = True     # ... and their dropout masks drawn on the device (False: by the host generator, as without batching)
        self.batched_streams = True          # the B streams of stateful training advanced together (streams.StreamBatcher)
        self.batched_streams_max_chars = 1 << 30
        self._engine_factory = engine_factory
        self._pool = None
        self._ctx_rows = None
        self._stop = False

    # ------------------------------------------------------------------ definition
    def configure(self):
        '''Define the model for the given parameters (rating.py:61-179).'''
        if self.stateful:
            self.variable_length = False
            self.first_window = 0
            self.batch_size = 1
        self.logger.info('using MI355X HIP implementation to compile %s %s model of depth %d width %d length %s size %d',
                         'stateful' if self.stateful else 'stateless',
                         'incremental' if self.incremental else 'contiguous',
                         self.depth, self.width,
                         'variable' if self.variable_length else str(self.length), self.voc_size)
        previous = self.model
        if self.voc_size > 0:
            factory = self._engine_factory
            if factory is None:
                from .engine import HipLM
                factory = HipLM
            self.model = factory(int(self.depth), int(self.width), int(self.voc_size), int(self.n_ctx))
            self.model.init_weights(seed=self.seed)
            # stateless contiguous graph (rating.py:98-99, 126-129): zero state per window, ONE output per window
            self.model.set_window_mode(not self.stateful and not self.incremental)
        else:
            self.model = None     # vocabulary unknown before the first training (rating.py:230)
        del previous
        self._pool = None
        self.status = 1

    def underspecify_contexts(self):
        '''Default input for context variables (rating.py:181-185).'''
        self.logger.info('using underspecification (zero) for %d context variables', self.n_ctx)
        return [0] * self.n_ctx

    # ------------------------------------------------------------------ training
    def train(self, data, val_data=None):
        '''Train model on text files (rating.py:248-310): stateful windows, one
        optimizer step per batch, validation after each epoch, early stopping with
        best-weights restore, per-epoch checkpoints, NaN abort, SIGINT graceful stop,
        state reset at file boundaries and before validation (callbacks.py:36-69).'''
        assert self.status > 0
        assert self.incremental is False
        if not self.stateful:
            return self._train_stateless(data, val_data)
        t_phase = time.perf_counter()
        self.timings = timings = {'split': 0.0, 'encode': 0.0, 'train_steps': 0.0, 'validation': 0.0, 'checkpoint': 0.0}
        (training_data, validation_data, _split, training_epoch_size, validation_epoch_size,
         total_size, steps) = self._split_data(data, val_data)
        timings['split'] = time.perf_counter() - t_phase
        self.logger.info('training on %d files / %d batches per epoch / %d character tokens for %d character types',
                         len(training_data), training_epoch_size, total_size, self.voc_size)
        self.reconfigure_for_mapping()
        lm = self.model
        if getattr(lm, "precision", PREC_BF16) != PREC_BF16:
            lm.prepare(PREC_BF16)
        sync = self._grad_sync()
        rank, world = sync.rank, sync.world
        sync.broadcast_params(lm)      # every rank initialised or loaded its own weights: rank 0's count
        B = max(1, int(self.streams))
        n_streams = B * world
        if len(training_data) < n_streams or len(validation_data) < 1:
            assert n_streams == 1 or len(training_data) >= n_streams, \
                "need at least %d training files for %d streams" % (n_streams, n_streams)
        rng = np.random.default_rng(self.seed)
        reset_rows = set()

        def make_streams(files, train):
            gens = []
            for s in range(B):
                gid = rank * B + s
                mine = files[gid::n_streams] or files[:1]

                def hook(name, s=s):
                    if train:
                        reset_rows.add(s)    # ResetStatesCallback.reset (callbacks.py:50-53)
                gens.append(windows.file_windows(mine, self.length, self.mapping[0], train=train, repeat=True, rng=rng,
                                                 on_new_file=hook, on_unmapped=self._unmapped_input,
                                                 char_degradation=self.char_degradation,
                                                 context_degradation=self.context_degradation))
            return gens

        def next_batch(gens):
            if isinstance(gens, streams.StreamBatcher):      # B streams advanced together (streams.py)
                batch, rows = gens.next_batch()
                if gens.train:
                    reset_rows.update(rows)                  # ResetStatesCallback.reset (callbacks.py:50-53)
                return batch
            xs, zs, ys = [], [], []
            for g in gens:
                x, z, y = next(g)
                xs.append(x); zs.append(z); ys.append(y)
            return np.stack(xs), np.stack(zs), np.stack(ys)

        def make_batcher(files, train):
            per_stream = [files[rank * B + s::n_streams] or files[:1] for s in range(B)]
            return streams.StreamBatcher(per_stream, self.length, self.mapping[0], train=train, rng=rng,
                                         char_degradation=self.char_degradation, context_degradation=self.context_degradation,
                                         on_unmapped=self._unmapped_input, device=getattr(lm, "device", None),
                                         codepoints=getattr(self, "_texts", None))

        # (the batched path keeps every file's ids in memory -- in HBM for the HIP engine --, 4 bytes per character;
        #  corpora beyond `batched_streams_max_chars` stay on the generator per stream, which re-reads file by file)
        t_phase = time.perf_counter()
        if self.batched_streams and total_size <= self.batched_streams_max_chars:
            train_gens = make_batcher(training_data, True)
            val_gens = make_batcher(validation_data, False)
            train_gens.prepare()
            val_gens.prepare()
            timings['encode'] = time.perf_counter() - t_phase
        else:
            train_gens = make_streams(training_data, True)
            val_gens = make_streams(validation_data, False)
        self._texts = {}
        draw_masks = getattr(lm, "draw_dropout_masks_device", None) if (self.batched_streams and self.device_dropout_masks) else None
        draw_masks = draw_masks or lm.draw_dropout_masks
        steps_per_epoch = max(1, ceil(training_epoch_size / n_streams))
        val_steps = max(1, ceil(validation_epoch_size / n_streams))
        history = {'loss': [], 'accuracy': [], 'val_loss': [], 'val_accuracy': []}
        best_val, best_weights, wait, stopped_epoch = None, None, 0, 0
        self._stop = False
        received = {'n': 0}

        def stopper(sig, _frame):      # StopSignalCallback (callbacks.py:6-34)
            if received['n']:
                self.logger.critical('interrupting')
                raise SystemExit(0)
            self.logger.critical('stopping training')
            received['n'] += 1
            self._stop = True
        try:
            old_handler = signal.signal(signal.SIGINT, stopper)
        except ValueError:             # not in the main thread
            old_handler = None
        nan_abort = False
        try:
            lm.reset_states(B)
            for epoch in range(self.max_epochs):
                t_phase = time.perf_counter()
                lm.read_loss(reset=True)
                loss_sum = acc_sum = 0.0
                # The launches are asynchronous: the next batch is generated on the host while the GPU
                # works on the current one, and only then is the loss read back (a synchronisation).
                # Streams that entered a new file while that batch was generated are reset right before
                # it is trained, as ResetStatesCallback.on_batch_begin does.
                # (the dropout masks of the next step are drawn there too: 2 ms of host work per 1024 streams)
                pending = (next_batch(train_gens), sorted(reset_rows), draw_masks(B))
                reset_rows.clear()
                for step in range(steps_per_epoch):
                    (x, z, y), rows, masks = pending
                    if rows:
                        lm.reset_states(B, rows=rows)
                    # (whatever fails on this rank -- a launch, a timed-out hand-off -- is kept until all ranks have agreed on
                    #  it below: the gradient all-reduce in between is entered by every rank, failed or not)
                    failure = None
                    try:
                        lm.train_window(x, z, y, masks)
                    except Exception as err:
                        failure = err
                    scale = sync.reduce(lm)
                    if failure is None:
                        try:
                            lm.adam_step(grad_scale=scale)
                        except Exception as err:
                            failure = err
                    if step + 1 < steps_per_epoch:
                        pending = (next_batch(train_gens), sorted(reset_rows), draw_masks(B))
                        reset_rows.clear()
                    ce, acc, reg = float('nan'), 0.0, 0.0
                    if failure is None:
                        try:
                            ce, acc, reg = lm.read_loss(reset=True)
                        except Exception as err:      # (a failed hand-off on this rank: every rank must leave together)
                            failure = err
                    loss = ce + reg
                    loss_sum += loss
                    acc_sum += acc
                    if loss > 25:
                        self.logger.warning('huge loss at batch %d', step)
                    # what ends the loop is decided by ALL ranks together: a rank that left on its own would leave
                    # the others waiting in the next all-reduce
                    is_nan, stop, failed = sync.any_flag(not np.isfinite(loss), self._stop, failure is not None)
                    if failed:
                        raise failure if failure is not None else RuntimeError('training failed on another rank')
                    if is_nan:    # TerminateOnNaN
                        self.logger.critical('NaN loss at batch %d', step)
                        nan_abort = True
                        break
                    if stop:
                        self._stop = True
                        break
                history['loss'].append(loss_sum / (step + 1))
                history['accuracy'].append(acc_sum / (step + 1))
                timings['train_steps'] += time.perf_counter() - t_phase
                if nan_abort:
                    break
                t_phase = time.perf_counter()
                # validation: states reset first (callbacks.py:67-69); dropout/regularisers off
                lm.reset_states(B)
                lm.prepare(PREC_BF16)
                v_loss = v_acc = 0.0
                nxt = next_batch(val_gens) if val_steps else None
                v_failure = None
                for k in range(val_steps):
                    x, z, y = nxt
                    try:
                        lm.forward_window(x, z, y, want_probs=False)
                    except Exception as err:
                        v_failure = v_failure or err
                    if k + 1 < val_steps:      # (generated while the GPU works, as in training)
                        nxt = next_batch(val_gens)
                    try:
                        ce, acc, _ = lm.read_loss(reset=True)
                    except Exception as err:
                        v_failure, ce, acc = v_failure or err, float('nan'), 0.0
                    v_loss += ce
                    v_acc += acc
                # (as in the training loop: a rank whose validation failed must not leave the others in the all-reduce below)
                (v_failed,) = sync.any_flag(v_failure is not None)
                if v_failed:
                    raise v_failure if v_failure is not None else RuntimeError('validation failed on another rank')
                v_loss, v_acc = sync.mean_scalars(v_loss / val_steps, v_acc / val_steps)
                history['val_loss'].append(v_loss)
                history['val_accuracy'].append(v_acc)
                lm.reset_states(B)
                timings['validation'] += time.perf_counter() - t_phase
                t_phase = time.perf_counter()
                self.logger.info('epoch %d: loss %.4f accuracy %.4f val_loss %.4f val_accuracy %.4f', epoch + 1,
                                 history['loss'][-1], history['accuracy'][-1], v_loss, v_acc)
                if best_val is None or v_loss < best_val:      # EarlyStopping / ModelCheckpoint
                    best_val, wait = v_loss, 0
                    best_weights = lm.get_weights()
                    if rank == 0:
                        self._checkpoint('ckpt.%02d-%.2f.h5' % (epoch + 1, v_loss))
                else:
                    wait += 1
                    if wait >= self.patience:
                        stopped_epoch = epoch
                        lm.set_weights(best_weights, PREC_BF16)
                        self.logger.info('early stopping at epoch %d, best weights restored', epoch + 1)
                        break
                timings['checkpoint'] += time.perf_counter() - t_phase
                if self._stop:
                    break
        finally:
            if old_handler is not None:
                signal.signal(signal.SIGINT, old_handler)
        self.history = history
        if history['val_loss']:
            self.logger.info('training finished with val_loss %f', min(history['val_loss']))
            if (np.isnan(history['val_loss'][-1]) or stopped_epoch == 0) and best_weights is not None:
                lm.set_weights(best_weights, PREC_BF16)     # rating.py:303-306
            self.status = 2
        else:
            self.logger.critical('training failed')
            self.status = 1

    def _grad_sync(self):
        from .distributed import GradSync
        return GradSync()

    def _checkpoint(self, filename):
        try:
            modelio.save_weights(filename, self.model.get_weights(), self.depth, self.n_ctx)
        except Exception as err:     # checkpoints are best effort
            self.logger.warning('cannot write checkpoint %s: %s', filename, err)

    def print_history(self):
        for k, v in self.history.items():
            print(f"{k}: {v}")

    def _split_data(self, data, val_data):
        '''Read text files, split into training vs validation, count batches and update
        the character mapping (stateful branch of rating.py:317-385).'''
        assert self.status >= 1
        # (the reference shuffles in place with the global `random`; a shuffled index list takes the same swaps, and
        # under data-parallel training every rank must see rank 0's order -- distributed.GradSync)
        order = list(range(len(data)))
        shuffle(order)
        order = self._grad_sync().broadcast_object(order)
        data[:] = [data[i] for i in order]
        if not self.stateful:
            return self._split_data_stateless(data, val_data)
        total_size = 0
        self._texts = {}
        chars = set(self.mapping[0].keys())
        steps = self.length
        if val_data:
            training_data, validation_data = data, val_data
        else:
            split = ceil(len(data) * self.validation_split)
            training_data, validation_data = data[:-split], data[-split:]
        assert training_data, "stateful mode needs at least one file for training"
        assert validation_data, "stateful mode needs at least one file for validation"
        for file in validation_data:
            self.logger.info('using input %s for validation only', file.name)
        sizes = []
        seen = np.zeros(windows.N_CODEPOINTS, dtype=bool)
        for group in (training_data, validation_data):
            epoch_size = 0
            for file in group:
                file.seek(0)
                text, size = windows.read_normalize_file(file)
                total_size += size
                epoch_size += ceil((size - self.length) / steps / self.batch_size)
                # (the distinct characters through a flag per code point: `set(text)` costs 2.5 ms per 50 k characters;
                #  the code point vectors are what train() maps to ids, it need not read the files again)
                cps = windows.codepoints(text)
                seen[cps] = True
                if self.batched_streams and total_size <= self.batched_streams_max_chars:
                    self._texts[id(file)] = cps
                else:
                    self._texts.clear()
            sizes.append(epoch_size)
        chars.update(chr(int(cp)) for cp in np.nonzero(seen)[0])
        chars = sorted(list(chars))
        self.voc_size = len(chars) + 1
        c_i = dict((c, i) for i, c in enumerate(chars, 1))
        i_c = dict((i, c) for i, c in enumerate(chars, 1))
        self.mapping = (c_i, i_c)
        return training_data, validation_data, None, sizes[0], sizes[1], total_size, steps

    def _split_data_stateless(self, data, val_data):
        '''window-wise split of the stateless mode (rating.py:352-378): windows advance by 3; without
        validation files both generators read the same data and share one uniform number per window
        position that assigns it to training or validation'''
        steps = 3
        total_size, max_size = 0, 0
        chars = set(self.mapping[0].keys())
        for file in data:
            file.seek(0)
            text, size = windows.read_normalize_file(file)
            total_size += size - self.length
            max_size = max(max_size, size)
            chars.update(set(text))
        if val_data:
            training_epoch_size = ceil(total_size / steps / self.batch_size)
            for file in val_data:
                file.seek(0)
                _text, size = windows.read_normalize_file(file)
                total_size += size - self.length
            validation_epoch_size = ceil(total_size / steps / self.batch_size)
            training_data, validation_data, split = data, val_data, None
        else:
            epoch_size = total_size / steps / self.batch_size
            training_epoch_size = ceil(epoch_size * (1 - self.validation_split))
            validation_epoch_size = ceil(epoch_size * self.validation_split)
            validation_data, training_data = data, data
            split = self._grad_sync().broadcast_object(np.random.uniform(0, 1, (ceil(max_size / steps),)))
        if self.first_window:
            training_epoch_size *= 1.0 + self.first_window
        chars = sorted(list(chars))
        self.voc_size = len(chars) + 1
        self.mapping = (dict((c, i) for i, c in enumerate(chars, 1)), dict((i, c) for i, c in enumerate(chars, 1)))
        return training_data, validation_data, split, training_epoch_size, validation_epoch_size, total_size, steps

    def _stateless_gen(self, files, steps, train, split=None, repeat=False):
        return windows.stateless_file_batches(
            files, self.length, self.mapping[0], steps, repeat=repeat, batch_size=self.batch_size, train=train, split=split,
            validation_split=self.validation_split, variable_length=self.variable_length, first_window=self.first_window,
            char_degradation=self.char_degradation, context_degradation=self.context_degradation,
            on_unmapped=self._unmapped_input)

    @staticmethod
    def _last_targets(x, y):
        tgt = np.full(x.shape, -1, dtype=np.int32)
        tgt[:, -1] = y
        return tgt

    def _train_stateless(self, data, val_data=None):
        '''rating.py:248-310 for stateful=False: batches of `batch_size` windows from zero state, one
        target per window; same epoch / early-stopping / checkpoint control as the stateful loop.
        (Single process: the window-wise split has no per-rank sharding.)'''
        (training_data, validation_data, split, training_epoch_size, validation_epoch_size,
         total_size, steps) = self._split_data(data, val_data)
        self.logger.info('training on %d files / %d batches per epoch / %d character tokens for %d character types',
                         len(training_data), training_epoch_size, total_size, self.voc_size)
        self.reconfigure_for_mapping()
        lm = self.model
        if getattr(lm, "precision", PREC_BF16) != PREC_BF16:
            lm.prepare(PREC_BF16)
        train_gen = self._stateless_gen(training_data, steps, True, split, repeat=True)
        val_gen = self._stateless_gen(validation_data, steps, False, split, repeat=True)
        steps_per_epoch = max(1, int(ceil(training_epoch_size)))
        val_steps = max(1, int(ceil(validation_epoch_size)))
        history = {'loss': [], 'accuracy': [], 'val_loss': [], 'val_accuracy': []}
        best_val, best_weights, wait, stopped_epoch = None, None, 0, 0
        nan_abort = False
        for epoch in range(self.max_epochs):
            lm.read_loss(reset=True)
            loss_sum = acc_sum = 0.0
            nxt = next(train_gen)
            for step in range(steps_per_epoch):
                x, z, y = nxt
                b = x.shape[0]
                lm.reset_states(b)
                one = lm.draw_dropout_masks(1)      # noise_shape (1, W): one mask for the whole batch (rating.py:150)
                lm.train_window(x, z, self._last_targets(x, y), np.repeat(one, b, axis=1))
                lm.adam_step()
                if step + 1 < steps_per_epoch:      # the next batch is generated while the GPU works (as in train)
                    nxt = next(train_gen)
                ce, acc, reg = lm.read_loss(reset=True)
                loss = ce + reg
                loss_sum += loss
                acc_sum += acc
                if not np.isfinite(loss):
                    self.logger.critical('NaN loss at batch %d', step)
                    nan_abort = True
                    break
            history['loss'].append(loss_sum / (step + 1))
            history['accuracy'].append(acc_sum / (step + 1))
            if nan_abort:
                break
            lm.prepare(PREC_BF16)
            v_loss = v_acc = 0.0
            n_rows = 0
            nxt = next(val_gen)
            for k in range(val_steps):
                x, z, y = nxt
                b = x.shape[0]
                lm.reset_states(b)
                lm.forward_window(x, z, self._last_targets(x, y), want_probs=False)
                if k + 1 < val_steps:
                    nxt = next(val_gen)
                ce, acc, _ = lm.read_loss(reset=True)
                v_loss += ce * b
                v_acc += acc * b
                n_rows += b
            v_loss, v_acc = v_loss / n_rows, v_acc / n_rows
            history['val_loss'].append(v_loss)
            history['val_accuracy'].append(v_acc)
            self.logger.info('epoch %d: loss %.4f accuracy %.4f val_loss %.4f val_accuracy %.4f', epoch + 1,
                             history['loss'][-1], history['accuracy'][-1], v_loss, v_acc)
            if best_val is None or v_loss < best_val:
                best_val, wait = v_loss, 0
                best_weights = lm.get_weights()
                self._checkpoint('ckpt.%02d-%.2f.h5' % (epoch + 1, v_loss))
            else:
                wait += 1
                if wait >= self.patience:
                    stopped_epoch = epoch
                    lm.set_weights(best_weights, PREC_BF16)
                    self.logger.info('early stopping at epoch %d, best weights restored', epoch + 1)
                    break
        self.history = history
        if history['val_loss']:
            self.logger.info('training finished with val_loss %f', min(history['val_loss']))
            if (np.isnan(history['val_loss'][-1]) or stopped_epoch == 0) and best_weights is not None:
                lm.set_weights(best_weights, PREC_BF16)
            self.status = 2
        else:
            self.logger.critical('training failed')
            self.status = 1

    def reconfigure_for_mapping(self):
        '''Reconfigure the character embedding after a change of mapping, transferring
        previous weights (rating.py:387-414).'''
        assert self.status >= 1
        old_voc = self.model.voc_size if self.model is not None else 0
        if old_voc < self.voc_size:
            if self.status >= 2 and self.model is not None:
                self.logger.warning('transferring weights from previous model with only %d character types', old_voc)
                old = self.model.get_weights()
                self.configure()
                new = self.model.get_weights()
                for name, value in old.items():
                    if name == 'E':
                        new['E'][0:old_voc, :] = value
                    else:
                        new[name] = value
                self.model.set_weights(new)
            else:
                self.configure()

    def remove_from_mapping(self, char=None, idx=None):
        '''Remove one character from mapping and shrink the embedding (rating.py:416-460).'''
        assert self.status > 1
        assert self.voc_size > 0
        if not char and not idx:
            return False
        if char:
            if char in self.mapping[0]:
                idx = self.mapping[0][char]
            else:
                self.logger.error('unmapped character "%s" cannot be removed', char)
                return False
        else:
            if idx in self.mapping[1]:
                char = self.mapping[1][idx]
            else:
                self.logger.error('unmapped index "%d" cannot be removed', idx)
                return False
        weights = self.model.get_weights()
        precision = getattr(self.model, "precision", PREC_SPLIT) or PREC_SPLIT
        self.logger.warning('pruning character "%s" [%d] with norm %f', char, idx, np.linalg.norm(weights['E'][idx, :]))
        c_i, i_c = self.mapping
        c_i.pop(char)
        i_c.pop(idx)
        for i in range(idx + 1, self.voc_size):
            other = i_c.pop(i)
            c_i[other] -= 1
            i_c[i - 1] = other
        self.voc_size -= 1
        weights['E'] = np.delete(weights['E'], idx, 0)
        self.configure()
        self.model.set_weights(weights, precision)
        self.status = 2
        return True

    # ------------------------------------------------------------------ windowed inference
    def _unmapped_input(self, char, position):
        self.logger.error('unmapped character "%s" at input position %d', char, position)

    def _ensure_precision(self):
        lm = self.model
        if getattr(lm, "precision", PREC_SPLIT) != PREC_SPLIT:
            lm.prepare(PREC_SPLIT)

    def test(self, test_data):
        '''Evaluate model on text files: exp(mean cross-entropy) (rating.py:462-491).'''
        assert self.status > 1
        assert self.incremental is False
        self._ensure_precision()
        lm = self.model
        if not self.stateful:
            # one window per character; the batch losses are averaged weighted by batch size, and only
            # ceil((size-1)/batch_size) generator batches per file are evaluated (rating.py:482-490)
            epoch_size = 0
            for file in test_data:
                file.seek(0)
                _text, size = windows.read_normalize_file(file)
                epoch_size += ceil((size - 1) / self.batch_size / 1)
            gen = self._stateless_gen(test_data, 1, False)
            lm.read_loss(reset=True)
            total, rows = 0.0, 0
            for _ in range(epoch_size):
                x, z, y = next(gen)
                lm.reset_states(x.shape[0])
                lm.forward_window(x, z, self._last_targets(x, y), want_probs=False)
                total += lm.read_loss(reset=True)[0] * x.shape[0]
                rows += x.shape[0]
            return exp(total / max(rows, 1))
        lm.reset_states(1)
        lm.read_loss(reset=True)
        n = 0
        total = 0.0
        for x, z, y in windows.file_windows(test_data, self.length, self.mapping[0], on_unmapped=self._unmapped_input):
            lm.forward_window(x[None], z[None], y[None], want_probs=False)
            n += 1
            if n % 64 == 0:
                total += lm.read_loss(reset=True)[0]
        total += lm.read_loss(reset=True)[0]
        return exp(total / max(n, 1))

    def rate(self, text, context=None):
        '''Rate a string all at once (rating.py:493-529): probability of every
        character given its predecessors; the first character gets 1.0.  The
        implicit LSTM state is NOT reset and is advanced over the zero-padded tail
        of the last window, exactly as the reference does.'''
        assert self.status > 1
        assert self.incremental is False
        if not context:
            context = self.underspecify_contexts()
        self._ensure_precision()
        text = windows.normalize(text)
        size = len(text)
        preds = []
        if not self.stateful:
            # Stateless scoring as the reference does it (rating.py:513-528): one window per character
            # position, of which only ceil((size-1)/batch_size) generator batches are evaluated -- the
            # partial windows of the first `length` positions count as batches of their own -- and
            # prediction k (made FOR character k) is paired with character k+1.  Both are reproduced.
            gen = windows.stateless_batches(text, context, self.length, self.mapping[0], 1, batch_size=self.batch_size,
                                            train=False, variable_length=self.variable_length,
                                            on_unmapped=self._unmapped_input)
            for _ in range(ceil((size - 1) / self.batch_size / 1)):
                x, z, _y = next(gen)
                self.model.reset_states(x.shape[0])
                preds.append(_np(self.model.forward_window(x, z))[:, -1])
        for x, z, _y in (windows.stateful_windows(text, context, self.length, self.mapping[0],
                                                  on_unmapped=self._unmapped_input) if self.stateful else ()):
            preds.append(_np(self.model.forward_window(x[None], z[None]))[0])
        probs = [1.0]
        if not preds:
            return probs[:size]
        preds = np.concatenate(preds, axis=0)[0:size]
        for pred, next_char in zip(preds, list(text[1:])):
            idx = self.mapping[0].get(next_char, 0)
            probs.append(pred[idx])
            if len(probs) >= size:
                break
        return probs

    def rate2(self, text, context=None):
        '''Rate a string one by one (rating.py:531-576): resets the state, feeds one
        character per step, returns [(char, prob)] and the perplexity 2^(H/len).'''
        assert self.status > 1
        assert self.incremental is False
        if not context:
            context = self.underspecify_contexts()
        context = windows.clamp_context(context)
        self._ensure_precision()
        text = windows.normalize(text)
        lm = self.model
        if not self.stateful:
            return self._rate2_stateless(text, context)
        lm.reset_states(1)
        z = np.asarray(context, dtype=np.int32).reshape(1, 1, -1)
        entropy = 0
        result = []
        prev = 0
        for i, char in enumerate(text):
            if char not in self.mapping[0]:
                self.logger.error('unmapped character "%s" at input position %d', char, i)
                idx = 0
            else:
                idx = self.mapping[0][char]
            if i == 0:
                result.append((char, 1.0))
            else:
                pred = _np(lm.forward_window(np.array([[prev]], dtype=np.int32), z))[0, 0]
                prob = float(pred[idx])
                entropy -= log(max(prob, 1e-99), 2)
                result.append((char, prob))
            prev = idx
        return result, pow(2.0, entropy / len(text))

    def _rate2_stateless(self, text, context):
        '''rating.py:548-576 for stateful=False: a window that slides by one character per step, filled from
        the right; evaluated over its last i columns (variable length) or whole, left zeros included'''
        lm = self.model
        L, n_ctx = self.length, len(context)
        x = np.zeros((1, L), dtype=np.int32)
        z = np.zeros((1, L, n_ctx), dtype=np.int32)
        entropy = 0
        result = []
        for i, char in enumerate(text):
            if char not in self.mapping[0]:
                self.logger.error('unmapped character "%s" at input position %d', char, i)
                idx = 0
            else:
                idx = self.mapping[0][char]
            if i == 0:
                result.append((char, 1.0))
            else:
                xi, zi = (x[:, -i:], z[:, -i:]) if self.variable_length else (x, z)
                lm.reset_states(1)
                pred = _np(lm.forward_window(np.ascontiguousarray(xi), np.ascontiguousarray(zi)))[0, -1]
                prob = float(pred[idx])
                entropy -= log(max(prob, 1e-99), 2)
                result.append((char, prob))
            x = np.roll(x, -1, axis=1)
            z = np.roll(z, -1, axis=1)
            x[0, -1] = idx
            z[0, -1] = np.asarray(context, dtype=np.int32)
        return result, pow(2.0, entropy / len(text))

    rate_once = rate2   # historic name of the one-by-one rater (north_star / BASELINE.json)

    # ------------------------------------------------------------------ incremental step
    def _state_pool(self):
        if self._pool is None:
            self._pool = StatePool(self.model, self.depth)
        return self._pool

    def _ids(self, candidates):
        c_i = self.mapping[0]
        return np.fromiter((c_i.get(c, 0) for c in candidates), dtype=np.int32, count=len(candidates))

    def _predict_refs(self, candidates, states, context, heads=False, targets=None):
        """device-resident variant of predict(): states are StateRef (or None = zero
        state); returns (probs [n,V] float32 array, list of new StateRef).  All index vectors travel to
        the GPU in ONE transfer; with `heads` the first `depth` vectors of every new state come back with
        the probabilities, so that history clustering (rating.py:887-916) compares them on the host
        instead of synchronising with the GPU once per candidate pair.

        An engine with `step_host` (HipLM: kl_step_batch_host) takes the index vectors as host arrays inside its launches
        and delivers into host memory without a stream synchronisation; with `targets` (the character ids a lattice
        decoder will look at, rating.py:838-843) it returns probs [n] -- those probabilities alone."""
        pool = self._state_pool()
        n = len(candidates)
        ctx = np.asarray(windows.clamp_context(context), dtype=np.int32)
        lm = self.model
        if hasattr(lm, "step_host"):
            new, slots = pool.refs(n)
            zero = pool.zero_slot
            c_i = self.mapping[0]
            packed = np.array([[c_i.get(c, 0) for c in candidates], [s.slot if s is not None else zero for s in states], slots],
                              dtype=np.int32)
            key = (n, tuple(ctx.tolist()))
            if self._ctx_rows is None or self._ctx_rows[0] != key:      # (one context per page: the same rows edge after edge)
                self._ctx_rows = (key, np.tile(ctx, (n, 1)))
            probs, hv = lm.step_host(packed[0], self._ctx_rows[1], packed[1], packed[2],
                                     target=targets, head_k=self.depth if heads else 0)
            if heads:
                for r, v in zip(new, hv):
                    r.head = v
            return probs, new
        new = [pool.ref() for _ in range(n)]
        packed = np.empty((3 + len(ctx), n), dtype=np.int32)
        packed[0] = self._ids(candidates)
        packed[1] = np.fromiter((s.slot if s is not None else pool.zero_slot for s in states), dtype=np.int32, count=n)
        packed[2] = np.fromiter((r.slot for r in new), dtype=np.int32, count=n)
        packed[3:] = ctx[:, None]
        hv = None
        if hasattr(lm, "to_device_i32"):
            dev = lm.to_device_i32(packed)
            ctx_d = dev[3] if len(ctx) == 1 else dev[3:].t().contiguous()
            if heads and hasattr(lm, "step_slots_heads"):
                probs, hv = lm.step_slots_heads(dev[0], ctx_d, dev[1], dev[2], self.depth)      # (one copy to the host for both)
            else:
                probs = _np(lm.step_slots(dev[0], ctx_d, dev[1], dev[2]))
        else:
            probs = _np(lm.step_slots(packed[0], packed[3:].T.copy(), packed[1], packed[2]))
        if heads:
            if hv is None:
                hv = _np(lm.pool_heads(packed[2], self.depth))
            for r, v in zip(new, hv):
                r.head = v
        return probs, new

    def predict(self, candidates, initial_states, context=None):
        '''Predict character probabilities for n hypotheses at once, passing initial
        and final states explicitly (rating.py:578-639).  `initial_states[i]` is None
        (zero state), a list [h1,c1,...,hL,cL] of arrays, or a StateRef.  Returns
        (list of n probability arrays [V], list of n state lists of (1,W) arrays).'''
        assert self.status > 1
        assert self.stateful is False
        assert self.incremental is True
        assert len(candidates) == len(initial_states), \
            "number of inputs (%d) and number of states (%d) inconsistent" % (len(candidates), len(initial_states))
        if not context:
            context = self.underspecify_contexts()
        self._ensure_precision()
        pool = self._state_pool()
        n = len(candidates)
        refs, upload_slots, upload_vals = [], [], []
        for state in initial_states:
            if state is None or (not isinstance(state, StateRef) and not state):
                refs.append(None)
            elif isinstance(state, StateRef):
                refs.append(state)
            else:
                ref = pool.ref()
                upload_slots.append(ref.slot)
                upload_vals.append(np.stack([np.asarray(s, dtype=np.float32).reshape(self.width) for s in state]))
                refs.append(ref)
        if upload_slots:
            self.model.pool_write(upload_slots, np.stack(upload_vals))
        probs, new = self._predict_refs(candidates, refs, context)
        states = self.model.pool_read([r.slot for r in new])      # [n][2L][W]
        preds = [probs[i, :] for i in range(n)]
        final_states = [[states[i, k][None, :] for k in range(2 * self.depth)] for i in range(n)]
        return preds, final_states

    # ------------------------------------------------------------------ beam searches
    def generate(self, prefix, length, context=None, variants=1):
        '''Generate `length` characters after `prefix` by beam search over the 10 best
        continuations with p >= 0.004 per hypothesis, 256 hypotheses wide
        (rating.py:642-709).  Returns `variants` strings, each starting with prefix[-1].'''
        assert self.status > 1
        assert self.stateful is False
        assert self.incremental is True
        if not context:
            context = self.underspecify_contexts()
        self._ensure_precision()
        state = None
        for char in prefix[:-1]:
            _, states = self._predict_refs([char], [state], context)
            state = states[0]
        next_fringe = [Node(state=state, value=prefix[-1], cost=0.0)]
        i_c = self.mapping[1]
        for _ in range(length):
            fringe = next_fringe
            preds, states = self._predict_refs([n.value for n in fringe], [n.state for n in fringe], context)
            # The reference insorts every continuation into one list ordered by cost and keeps the first
            # 256 (rating.py:699-707).  Same result, less Python: the order is kept on a parallel list of
            # float keys (insort_left = bisect_left, ties go in front), and a continuation that is
            # strictly worse than the current 256th can never return into the kept part, so it is
            # not even built.
            next_fringe, keys = [], []
            for j, n in enumerate(fringe):
                pred = preds[j]
                pred_best = np.argsort(pred)[-10:]
                pred_best = pred_best[np.searchsorted(pred[pred_best], 0.004):]
                costs = -np.log(pred[pred_best])
                state = states[j]
                base = n.cum_cost
                for best, cost in zip(pred_best, costs):
                    if best not in i_c:
                        continue
                    if len(keys) >= 256 and base + cost > keys[-1]:
                        continue
                    node = Node(parent=n, state=state, value=i_c[best], cost=cost)
                    pos = bisect_left(keys, node.cum_cost)
                    keys.insert(pos, node.cum_cost)
                    next_fringe.insert(pos, node)
                    if len(keys) > 256:
                        keys.pop()
                        next_fringe.pop()
        best = next_fringe[0:variants]
        return [''.join([n.value for n in res.to_sequence()]) for res in best]

    def rate_best(self, graph, start_node, end_node, start_traceback=None, context=None, lm_weight=0.5,
                  beam_width=10, beam_clustering_dist=0):
        '''Rate a lattice of string alternatives, decoding the best-scoring path
        incrementally (rating.py:712-859).  `graph` is a networkx.DiGraph whose edges
        carry `element` and `alternatives` (objects with `.Unicode`, `.conf`, `.index`).
        Returns (path [(element, alternative, score)], entropy, traceback).

        The beam bookkeeping lives in lattice_beam.py: one table of tracks per edge, list operations on track
        numbers and float keys, tree nodes only for the hypotheses that survive at an edge's destination.'''
        if not context:
            context = self.underspecify_contexts()
        self._ensure_precision()
        if not start_traceback:
            root = Node(state=None, value='\n', cost=0.0)
            start_traceback = ([root], root)
        graph.nodes[start_node]['traceback'] = start_traceback[0]
        clustering = bool(beam_clustering_dist)

        def predict(chars, states, targets=None):
            return self._predict_refs(chars, states, context, heads=clustering, targets=targets)

        def close_states(a, b):
            return all(self._state_distance_below(a, b, k, beam_clustering_dist) for k in range(self.depth))

        reached = None
        for source, reached in lattice_beam.lattice_edges(graph, start_node):
            edge = graph.edges[source, reached]
            element, alternatives = edge['element'], edge['alternatives']
            self.logger.debug("rating %d alternatives from %d to %d", len(alternatives), source, reached)
            assert 'traceback' in graph.nodes[source], \
                "breadth-first search should have visited %d first in '%s'" % (source, element.id)
            target = graph.nodes[reached]
            tracks = lattice_beam.EdgeTracks(graph.nodes[source]['traceback'], alternatives, element, self.mapping[0],
                                             lm_weight, self.logger)
            finished = lattice_beam.FinishedBeam(target.get('traceback', []))
            lattice_beam.decode_edge(tracks, finished, predict, self.batch_size,
                                     max(len(a.Unicode) for a in alternatives) * 3,
                                     close_states if clustering else None)
            target['traceback'] = [ref if kind == "node" else tracks.node(ref) for kind, ref in finished.items[:beam_width]]
        assert reached == end_node, \
            'breadth-first search failed to reach true end node (%s instead of %d)' % (reached, end_node)
        assert 'traceback' in graph.nodes[reached], \
            "breadth-first search failed to reach end node with any result"
        return self.next_path(graph.nodes[reached]['traceback'], start_traceback)

    def next_path(self, beam, traceback):
        '''Advance from `traceback` to `beam` (rating.py:862-885): lock into the best
        hypothesis' ancestor in the previous beam, emit its path, cut the others.'''
        return lattice_beam.advance_traceback(beam, traceback)

    def _state_distance_below(self, a, b, k, distance):
        if isinstance(a, StateRef) and isinstance(b, StateRef):
            if a.head is not None and b.head is not None:
                d = a.head[k] - b.head[k]
                return float(np.dot(d, d)) < distance * distance
            d2 = float(_np(self.model.state_dist2([a.slot], [b.slot], k))[0])
            return d2 < distance * distance
        return np.linalg.norm(np.asarray(a[k]) - np.asarray(b[k])) < distance

    # ------------------------------------------------------------------ model I/O
    def save(self, filename):
        '''Save weights and configuration (rating.py:918-945).'''
        assert self.status > 1
        config = {
            'history': json.dumps(self.history, cls=modelio.NumpyEncoder),
            'width': int(self.width), 'depth': int(self.depth), 'length': int(self.length),
            'stateful': bool(self.stateful), 'variable_length': bool(self.variable_length),
            'mapping': np.fromiter((ord(self.mapping[1][i]) if i in self.mapping[1] else 0
                                    for i in range(self.voc_size)), dtype='uint32'),
        }
        modelio.save_model(filename, self.model.get_weights(), config, self.depth, self.n_ctx)

    def load_config(self, filename):
        '''Load parameters to prepare configuration (rating.py:947-964).'''
        assert self.status == 0
        config = modelio.load_config(filename)
        history = config.get('history')
        self.history = json.loads(history) if history else {}
        self.width = int(config['width'])
        self.depth = int(config['depth'])
        self.length = int(config['length'])
        self.stateful = bool(config['stateful'])
        self.variable_length = bool(config['variable_length'])
        mapping = config['mapping']
        c_i = dict((chr(c), i) for i, c in enumerate(mapping) if c > 0)
        i_c = dict((i, chr(c)) for i, c in enumerate(mapping) if c > 0)
        self.mapping = (c_i, i_c)
        self.voc_size = len(c_i) + 1

    def load_weights(self, filename):
        '''Load weights into the configured model (rating.py:966-974).'''
        assert self.status > 0
        weights = modelio.load_weights(filename, self.depth, self.width, self.n_ctx)
        self.model.set_weights(weights, PREC_SPLIT)      # (rating needs the f32-accurate mode; train() re-prepares in bf16)
        self.status = 2

    # ---- offline views of the embeddings (rating.py:1169-1237; matplotlib / scikit-learn are imported on use)
    def _embedding(self, name):
        assert self.status == 2
        weights = self.model.get_weights()
        if name not in weights:
            raise KeyError("the model has no embedding '%s'" % name)
        return np.asarray(weights[name], dtype=np.float64)

    def plot_char_embeddings_similarity(self, filename):
        '''Paint a heat map of character embeddings: |E E^T| as a grayscale PNG (rating.py:1169-1187).'''
        import logging
        logging.getLogger('matplotlib').setLevel(logging.WARNING)
        from matplotlib import pyplot as plt
        from matplotlib import cm
        charwgt = self._embedding('E')
        plt.imsave(filename, np.abs(np.dot(charwgt, charwgt.T)), cmap=cm.gray)

    def plot_context_embeddings_similarity(self, filename, n=1):
        '''Paint a heat map of the n-th context variable's embeddings (rating.py:1189-1207).'''
        import logging
        logging.getLogger('matplotlib').setLevel(logging.WARNING)
        from matplotlib import pyplot as plt
        from matplotlib import cm
        ctxtwgt = self._embedding('Ctx%d' % (n - 1))
        plt.imsave(filename, np.abs(np.dot(ctxtwgt, ctxtwgt.T)), cmap=cm.gray)

    def plot_context_embeddings_projection(self, filename, n=1):
        '''Scatter plot of a 2-d PCA projection of the n-th context variable's embeddings, every point
        labelled with its decade (rating.py:1209-1237).  Labels are de-overlapped when adjustText is there.'''
        import logging
        logging.getLogger('matplotlib').setLevel(logging.WARNING)
        import matplotlib
        matplotlib.use('Agg')
        from matplotlib import pyplot as plt
        from sklearn.decomposition import PCA
        ctxtprj = PCA(n_components=2).fit_transform(self._embedding('Ctx%d' % (n - 1)))
        plt.figure(figsize=(11.7, 8.3))
        plt.plot(ctxtprj[:, 0], ctxtprj[:, 1], 'bo', markersize=2)
        texts = [plt.text(xy[0], xy[1], str(year) + 'x', c='b', size='xx-small') for year, xy in enumerate(ctxtprj)]
        try:
            from adjustText import adjust_text
            adjust_text(texts, time_lim=20, iter_lim=20, expand_axes=True, arrowprops=dict(arrowstyle="-", color='b', lw=0.5))
        except ImportError:
            pass
        plt.tick_params(left=False, right=False, bottom=False, labelleft=False, labelbottom=False)
        plt.savefig(filename)
        plt.close()

    def print_charset(self):
        '''Print the mapped characters (rating.py:1160-1167).'''
        import unicodedata
        for i, c in self.mapping[1].items():
            print('%d: "%s"' % (i, c))
            char = unicodedata.normalize('NFC', c)
            if c != char:
                self.logger.warning('mapped character "%s" (%d) should have been normalized to "%s", which is %s mapped',
                                    c, i, char, 'also' if char in self.mapping[0] else 'not')
