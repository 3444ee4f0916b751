"""HipLM: the MI355X engine behind `Rater.model`.

Plays the role the compiled Keras model plays in the reference
(ocrd_keraslm/lib/rating.py:56, 171-178): it owns the weights, the optimizer
moments, the implicit LSTM state of stateful streams and the state pool of
incremental hypotheses -- all as torch tensors in HBM -- and runs every
contraction through the hand-written gfx950 kernels behind the C ABI
(include/keraslm_hip.h).  torch is used for device memory, streams and (in
`distributed.py`) the RCCL all-reduce only.  There is no CPU fallback.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math

import numpy as np

from . import hipabi

CTX_VOCAB = 200   # rating.py:111
CTX_DIM = 10
DROPOUT_RATE = 0.1  # rating.py:152


FAST_WIDTHS = (64, 128, 256, 512, 1024)   # hidden sizes the persistent scans are instantiated for


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def physical_width(width):
    """The width the kernels run at: the next one the persistent scans serve (else the next multiple of 32).
    The extra hidden units are zero-padding -- all their weights and biases are zero, so their cell and
    output stay exactly zero, nothing flows out of them, and their gradients are exactly zero (Adam leaves
    them alone): a model of ANY width (the reference takes 1..9128, scripts/run.py:34) runs bit-identically to
    its padded twin, on the fast kernels."""
    for w in FAST_WIDTHS:
        if width <= w:
            return w
    return (width + 31) // 32 * 32


def _solve_lower(low, rhs):
    """low^-1 rhs for a lower triangular matrix"""
    from scipy.linalg import solve_triangular
    return solve_triangular(low, rhs, lower=True, check_finite=False)


class HipLM:
    def __init__(self, depth, width, voc_size, n_ctx=1, device="cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise hipabi.KlError("no GPU visible: the Rater hot path runs on MI355X only (no CPU fallback)")
        self.torch = torch
        self.lib = hipabi.load()
        self.device = torch.device(device)
        self.depth, self.width, self.voc_size, self.n_ctx = int(depth), int(width), int(voc_size), int(n_ctx)
        if self.width < 1:
            raise hipabi.KlError("width must be positive (got %d)" % self.width)
        self.pwidth = physical_width(self.width)       # what the kernels see (zero-padded hidden units)
        self.padded = self.pwidth != self.width
        self.cfg = hipabi.KlConfig(self.depth, self.pwidth, self.voc_size, self.n_ctx, CTX_VOCAB, CTX_DIM)
        self.handle = self.lib.kl_create(C.byref(self.cfg))
        if not self.handle:
            raise hipabi.KlError("kl_create rejected configuration depth=%d width=%d voc=%d"
                                 % (self.depth, self.width, self.voc_size))
        self.n_params = self.lib.kl_param_count(C.byref(self.cfg))
        self.layout = self._read_layout()
        # all launches go to one side stream: whole windows are replayed as hipGraphs,
        # and the legacy default stream cannot be captured
        self.stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.device(self.device):
            self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            nbytes = self.lib.kl_derived_bytes(self.handle)
            self.derived = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            hipabi.check(self.lib.kl_bind(self.handle, _ptr(self.params), _ptr(self.derived), nbytes), "kl_bind")
        self.precision = 0
        self.grads = None
        self.adam_m = None
        self.adam_v = None
        self.adam_t = 0
        self.loss_acc = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._ws = None
        self._ws_key = None
        self.states = None          # [B][2L][W] implicit state of the stateful streams
        self.pool = None            # [slots][2L][W] explicit states of hypotheses
        self.max_streams_per_launch = 0      # 0: what the kernels address (train_window splits larger batches into groups)
        self._part_grads = None
        self._part_loss = None
        self._pad_states = {}
        self.pad_streams = True              # pad a group of streams up to the next fast count (HipLM._padded_streams)
        self.plan_override = None            # width 512: a stream plan to use instead of _plan_512's (tests)
        self._step_ws = None
        self._step_host_ws = None
        self._hio = None
        self._step_ws_bytes = {}
        self.last_only = False
        self._rng = np.random.default_rng(0)

    def __del__(self):
        try:
            if getattr(self, "_hio", None) is not None:
                self._host_io_free()
            if getattr(self, "handle", None):
                self.lib.kl_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def _read_layout(self):
        out = []
        i = 0
        name = C.create_string_buffer(32)
        off, rows, cols = C.c_size_t(), C.c_size_t(), C.c_size_t()
        while self.lib.kl_param_layout(C.byref(self.cfg), i, name, 32, C.byref(off), C.byref(rows), C.byref(cols)) == 0:
            out.append((name.value.decode(), off.value, rows.value, cols.value))
            i += 1
        return out

    def _stream(self):
        return C.c_void_p(self.stream.cuda_stream)

    @contextlib.contextmanager
    def _launch(self):
        """Run the enclosed launches on the engine stream, ordered after what the
        caller's current stream has enqueued and before what it enqueues next."""
        torch = self.torch
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        try:
            with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
                yield
        finally:
            cur.wait_stream(self.stream)      # (also when the body raised: the streams stay joined)

    # `layout`, `params`, `grads`, `states`, `pool` are PHYSICAL (padded width); the accessors below speak the
    # model's own shapes.  For the widths in FAST_WIDTHS (and multiples of 32 above 1024) both coincide.
    def _logical_shape(self, name):
        W = self.width
        if name == "E":
            return (self.voc_size, W)
        if name.startswith("Ctx"):
            return (CTX_VOCAB, CTX_DIM)
        if name.startswith("b"):
            return (4 * W,)
        if name == "K0":
            return (W + self.n_ctx * CTX_DIM, 4 * W)
        return (W, 4 * W)

    def _row_map(self, name, n_rows):
        """physical row of every logical row (K0: the context rows sit behind the padded embedding rows)"""
        r = np.arange(n_rows)
        if name == "K0":
            r = np.where(r < self.width, r, r + (self.pwidth - self.width))
        return r

    def _pad(self, name, a):
        """logical array -> physical array (zeros in the padding)"""
        a = np.asarray(a, dtype=np.float32).reshape(self._logical_shape(name))
        if not self.padded or name.startswith("Ctx"):
            return a
        W, Wp = self.width, self.pwidth
        if name == "E":
            out = np.zeros((a.shape[0], Wp), dtype=np.float32)
            out[:, :W] = a
            return out
        if name.startswith("b"):
            out = np.zeros(4 * Wp, dtype=np.float32)
            for g in range(4):
                out[g * Wp:g * Wp + W] = a[g * W:(g + 1) * W]
            return out
        rows = self._row_map(name, a.shape[0])
        out = np.zeros((a.shape[0] + (Wp - W), 4 * Wp), dtype=np.float32)
        for g in range(4):
            out[rows, g * Wp:g * Wp + W] = a[:, g * W:(g + 1) * W]
        return out

    def _unpad(self, name, a):
        """physical array -> logical array"""
        if not self.padded or name.startswith("Ctx"):
            return a
        W, Wp = self.width, self.pwidth
        if name == "E":
            return a[:, :W].copy()
        if name.startswith("b"):
            return np.concatenate([a[g * Wp:g * Wp + W] for g in range(4)])
        rows = self._row_map(name, self._logical_shape(name)[0])
        return np.concatenate([a[rows, g * Wp:g * Wp + W] for g in range(4)], axis=1)

    def _unflatten(self, flat):
        out = {}
        for name, off, rows, cols in self.layout:
            a = flat[off:off + rows * cols]
            a = a.reshape(cols).copy() if name.startswith("b") else a.reshape(rows, cols).copy()
            out[name] = self._unpad(name, a)
        return out

    def get_weights(self):
        """dict name -> float32 array in Keras shapes (E, Ctx0.., K0, U0, b0, ...)."""
        return self._unflatten(self.params.detach().cpu().numpy())

    def get_grads(self):
        """gradients of the last train_window, same names and shapes as get_weights()"""
        return self._unflatten(self.grads.detach().cpu().numpy())

    def set_weights(self, weights, precision=None):
        flat = np.empty(self.n_params, dtype=np.float32)
        for name, off, rows, cols in self.layout:
            a = np.asarray(weights[name], dtype=np.float32)
            if a.size != int(np.prod(self._logical_shape(name))):
                raise ValueError("weight %s has %d elements, expected shape %s" % (name, a.size, self._logical_shape(name)))
            flat[off:off + rows * cols] = self._pad(name, a).reshape(-1)
        with self._launch():
            self.params.copy_(self.torch.from_numpy(flat))
        self.prepare(precision or self.precision or hipabi.KL_PREC_SPLIT)

    def get_states(self):
        """implicit states of the stateful streams, numpy [B][2L][W]"""
        return self.states[:, :, :self.width].cpu().numpy()

    def set_states(self, values):
        values = np.asarray(values, dtype=np.float32)
        self.reset_states(values.shape[0])
        with self._launch():
            self.states[:, :, :self.width] = self.torch.from_numpy(np.ascontiguousarray(values)).to(self.device)

    def init_weights(self, seed=None, emb_std=0.001):
        """Keras initialisers of rating.py:104-114 + LSTM defaults (glorot_uniform kernel,
        orthogonal recurrent kernel, zero bias with unit forget gate)."""
        rng = np.random.default_rng(seed)
        W = self.width
        w = {}
        for name, _off, _rows, _cols in self.layout:
            shape = self._logical_shape(name)
            rows, cols = (1, shape[0]) if len(shape) == 1 else shape
            if name == "E" or name.startswith("Ctx"):
                w[name] = (rng.standard_normal((rows, cols)) * emb_std).astype(np.float32)
            elif name.startswith("K"):
                lim = math.sqrt(6.0 / (rows + cols))
                w[name] = rng.uniform(-lim, lim, (rows, cols)).astype(np.float32)
            elif name.startswith("U"):
                # Keras' Orthogonal: the Q of a QR factorisation of a [4W, W] Gaussian matrix, signs fixed so that R's diagonal
                # is positive.  That factorisation is unique, and for such a well-conditioned matrix (condition number 3) it is
                # had from a Cholesky factor of the Gram matrix -- matrix products instead of Householder steps (0.28 s each
                # at width 512).  Done twice, the second pass removes the rounding of the first.
                a = rng.standard_normal((cols, rows))
                qt = a.T
                for _ in range(2):
                    low = np.linalg.cholesky(qt @ qt.T)
                    qt = _solve_lower(low, qt)
                w[name] = qt.astype(np.float32)
            else:
                b = np.zeros(cols, dtype=np.float32)
                b[W:2 * W] = 1.0
                w[name] = b
        self.set_weights(w, self.precision or hipabi.KL_PREC_SPLIT)

    def prepare(self, precision):
        with self._launch():
            hipabi.check(self.lib.kl_prepare(self.handle, int(precision), self._stream()), "kl_prepare")
        self.precision = int(precision)

    # ------------------------------------------------------------------ windows
    def _workspace(self, B, T, training):
        key = (B, T, bool(training))
        if self._ws_key != key:
            n = self.lib.kl_window_workspace_bytes(self.handle, B, T, 1 if training else 0)
            self._ws = None
            self._ws = self.torch.empty(n, dtype=self.torch.uint8, device=self.device)
            self._ws_key = key
        return self._ws

    def set_window_mode(self, last_only):
        """False (default): the stateful graph -- a target at every position, means over B*T positions.
        True: the stateless graph (rating.py:126-129) -- one target per window at its last position,
        means over the B windows."""
        hipabi.check(self.lib.kl_set_window_mode(self.handle, 1 if last_only else 0), "kl_set_window_mode")
        self.last_only = bool(last_only)

    def reset_states(self, B=None, rows=None):
        """Keras reset_states (rating.py:475, 555; callbacks.py:58, 69)."""
        if self.states is None or (B is not None and self.states.shape[0] != B):
            self.states = self.torch.zeros((B or 1, 2 * self.depth, self.pwidth), dtype=self.torch.float32,
                                           device=self.device)
        elif rows is None:
            self.states.zero_()
        else:
            self.states[self.torch.as_tensor(rows, device=self.device, dtype=self.torch.long)] = 0

    def _dev_i32(self, a):
        if isinstance(a, self.torch.Tensor):
            return a.to(device=self.device, dtype=self.torch.int32).contiguous()
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def forward_window(self, idx, ctx, tgt=None, want_probs=True):
        """idx [B,T], ctx [B,T,n_ctx], tgt [B,T] (-1 = padded) or None.
        Returns probs [B,T,V] (numpy) or None; with tgt also accumulates loss_acc."""
        torch = self.torch
        if self.precision == 0:
            raise hipabi.KlError("weights not prepared")
        with self._launch():
            idx_d = self._dev_i32(idx)
            B, T = idx_d.shape
            ctx_d = self._dev_i32(ctx) if self.n_ctx else None
            tgt_d = self._dev_i32(tgt) if tgt is not None else None
            if self.states is None or self.states.shape[0] != B:
                self.reset_states(B)
            # (bf16 precision = validation windows: a training-size workspace lets the library take the training
            # forward, i.e. the persistent scans; it is the buffer train_window uses anyway)
            training_ws = self.precision == hipabi.KL_PREC_BF16
            probs = torch.empty((B, T, self.voc_size), dtype=torch.float32, device=self.device) if want_probs else None
            parts = self._stream_groups(B, T) if training_ws else self._rating_groups(B)
            # (validation windows -- bf16, a loss and no probabilities wanted -- are padded like training batches)
            padded = [self._padded_streams(b1 - b0, T) if (training_ws and not want_probs and tgt_d is not None) else b1 - b0
                      for b0, b1 in parts]
            if len(parts) == 1 and padded[0] == B:
                ws = self._workspace(B, T, training_ws)
                hipabi.check(self.lib.kl_forward_window(self.handle, B, T, _ptr(idx_d), _ptr(ctx_d), _ptr(tgt_d),
                                                        _ptr(self.states), _ptr(probs), _ptr(self.loss_acc), _ptr(ws),
                                                        ws.numel(), self._stream()), "kl_forward_window")
            else:
                # (more streams than one launch sequence addresses: groups of streams one after the other, as in train_window;
                # the means over the batch are the size-weighted sums of the groups' means)
                if self._part_loss is None:
                    self._part_loss = torch.zeros_like(self.loss_acc)
                ws = self._workspace(max(padded), T, training_ws)
                for (b0, b1), Bp in zip(parts, padded):
                    n = b1 - b0
                    self._part_loss.zero_()
                    x, c, st = idx_d[b0:b1], (ctx_d[b0:b1] if ctx_d is not None else None), self.states[b0:b1]
                    y = tgt_d[b0:b1] if tgt_d is not None else None
                    if Bp != n:
                        x = torch.nn.functional.pad(x, (0, 0, 0, Bp - n))
                        if y is not None:      # (-2, not the -1 of a padded tail: a dummy stream is no hit for the accuracy either)
                            y = torch.nn.functional.pad(y, (0, 0, 0, Bp - n), value=-2)
                        if c is not None:
                            c = torch.nn.functional.pad(c, (0, 0) * (c.dim() - 1) + (0, Bp - n))
                        st = self._pad_states.get(Bp)
                        if st is None:
                            st = self._pad_states[Bp] = torch.zeros((Bp,) + tuple(self.states.shape[1:]), dtype=torch.float32,
                                                                    device=self.device)
                        st[:n] = self.states[b0:b1]
                        st[n:] = 0
                        hipabi.check(self.lib.kl_set_loss_rows(self.handle, n), "kl_set_loss_rows")
                    try:
                        hipabi.check(self.lib.kl_forward_window(self.handle, Bp, T, _ptr(x), _ptr(c), _ptr(y), _ptr(st),
                                                                _ptr(probs[b0:b1] if probs is not None else None),
                                                                _ptr(self._part_loss), _ptr(ws), ws.numel(), self._stream()),
                                     "kl_forward_window")
                    finally:
                        if Bp != n:
                            hipabi.check(self.lib.kl_set_loss_rows(self.handle, 0), "kl_set_loss_rows")
                    if Bp != n:
                        self.states[b0:b1] = st[:n]
                    self.loss_acc[:2] += (n / B) * self._part_loss[:2]
                    self.loss_acc[3] = torch.maximum(self.loss_acc[3], self._part_loss[3])
            if want_probs and float(self.loss_acc[3].item()) != 0.0:
                # (the caller reads the probabilities next, so this sync is not an extra one)
                self.loss_acc[3] = 0.0        # one timed-out window must not fail every later call
                raise hipabi.KlError("persistent scan hand-off timed out (kl_forward_window)")
        return probs

    def draw_dropout_masks(self, B):
        """Inverted-dropout keep masks [L][B][W], time-constant per window (rating.py:146-152)."""
        keep = self._rng.random((self.depth, B, self.width)) >= DROPOUT_RATE
        m = (keep / (1.0 - DROPOUT_RATE)).astype(np.float32)
        m[0] = 1.0
        return m

    def draw_dropout_masks_device(self, B):
        """The same masks drawn in HBM (torch's generator, seeded from this engine's numpy generator on first use): at 3072
        streams the host drew 3.1 M random numbers per step, 6-8 ms beside 26 ms of GPU work."""
        torch = self.torch
        if getattr(self, "_tgen", None) is None:
            self._tgen = torch.Generator(device=self.device)
            self._tgen.manual_seed(int(self._rng.integers(0, 2 ** 31 - 1)))
        with torch.cuda.device(self.device):
            keep = torch.rand((self.depth, B, self.pwidth), device=self.device, generator=self._tgen) >= DROPOUT_RATE
            m = keep.to(torch.float32) / (1.0 - DROPOUT_RATE)
            m[0] = 1.0
        return m

    def ensure_training_buffers(self):
        torch = self.torch
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
            self.adam_t = 0

    def train_window(self, idx, ctx, tgt, masks=None):
        """forward + backward of one batch of B stateful windows; fills self.grads."""
        torch = self.torch
        self.ensure_training_buffers()
        if self.precision != hipabi.KL_PREC_BF16:
            self.prepare(hipabi.KL_PREC_BF16)
        with self._launch():
            idx_d = self._dev_i32(idx)
            B, T = idx_d.shape
            ctx_d = self._dev_i32(ctx) if self.n_ctx else None
            tgt_d = self._dev_i32(tgt)
            if self.states is None or self.states.shape[0] != B:
                self.reset_states(B)
            masks_d = None
            if masks is not None:
                masks_d = masks if isinstance(masks, torch.Tensor) else torch.from_numpy(
                    np.ascontiguousarray(masks, dtype=np.float32)).to(self.device)
                if self.padded and masks_d.shape[-1] == self.width:      # (whatever the padded units get is multiplied by zero)
                    masks_d = torch.nn.functional.pad(masks_d, (0, self.pwidth - self.width), value=1.0).contiguous()
            parts = self._stream_groups(B, T)
            padded = [self._padded_streams(b1 - b0, T) for b0, b1 in parts]
            if len(parts) == 1 and padded[0] == B:
                ws = self._workspace(B, T, True)
                hipabi.check(self.lib.kl_train_window(self.handle, B, T, _ptr(idx_d), _ptr(ctx_d), _ptr(tgt_d),
                                                      _ptr(self.states), _ptr(masks_d), _ptr(self.grads),
                                                      _ptr(self.loss_acc), _ptr(ws), ws.numel(), self._stream()),
                             "kl_train_window")
                return
            # More streams than one launch sequence addresses (the scans index a layer's gate rows, T * B * 4W bf16, with 32-bit
            # buffer offsets: 3072 streams at cfg2), or a count none of the fast kernels takes: the streams are independent, so
            # the batch runs as groups of streams one after the other -- each on the persistent-scan path, a group that is just
            # short of a fast count PADDED with dummy streams (targets -1: no loss, no gradient; kl_set_loss_rows keeps the
            # means those over the real streams) -- and the gradient of the mean over all B*T positions is the size-weighted sum
            # of the groups' gradients (the regularisers' part is the same in every group: the weights sum to 1).
            if self._part_grads is None:
                self._part_grads = torch.zeros_like(self.grads)
            if self._part_loss is None:
                self._part_loss = torch.zeros_like(self.loss_acc)
            ws = self._workspace(max(padded), T, True)
            for i, ((b0, b1), Bp) in enumerate(zip(parts, padded)):
                n = b1 - b0
                wgt = n / B
                self._part_loss.zero_()
                m = masks_d[:, b0:b1] if masks_d is not None else None
                c = ctx_d[b0:b1] if ctx_d is not None else None
                x, y, st = idx_d[b0:b1], tgt_d[b0:b1], self.states[b0:b1]
                if Bp != n:
                    x = torch.nn.functional.pad(x, (0, 0, 0, Bp - n))
                    # (-2, not the -1 of a padded tail: Keras counts an all-zero target row as a hit when class 0 wins, a dummy stream
                    #  must not be counted at all)
                    y = torch.nn.functional.pad(y, (0, 0, 0, Bp - n), value=-2)
                    if c is not None:
                        c = torch.nn.functional.pad(c, (0, 0) * (c.dim() - 1) + (0, Bp - n))
                    if m is not None:
                        # (the dummy streams' keep-masks repeat real streams': the register-tile backward scan keeps a mask as
                        #  one bit per cell plus ONE scale per thread and refuses a mask with a second non-zero value)
                        m = torch.cat([m, m[:, :Bp - n]], dim=1)
                    st = self._pad_states.get(Bp)
                    if st is None:      # (one buffer per padded count: its address is part of the replayed launch sequence's key)
                        st = self._pad_states[Bp] = torch.zeros((Bp,) + tuple(self.states.shape[1:]), dtype=torch.float32,
                                                                device=self.device)
                    st[:n] = self.states[b0:b1]
                    st[n:] = 0
                    hipabi.check(self.lib.kl_set_loss_rows(self.handle, n), "kl_set_loss_rows")
                if m is not None:
                    m = m.contiguous()
                try:
                    hipabi.check(self.lib.kl_train_window(self.handle, Bp, T, _ptr(x), _ptr(c), _ptr(y), _ptr(st), _ptr(m),
                                                          _ptr(self._part_grads), _ptr(self._part_loss), _ptr(ws), ws.numel(),
                                                          self._stream()), "kl_train_window")
                finally:
                    if Bp != n:
                        hipabi.check(self.lib.kl_set_loss_rows(self.handle, 0), "kl_set_loss_rows")
                if Bp != n:
                    self.states[b0:b1] = st[:n]
                if i == 0:
                    torch.mul(self._part_grads, wgt, out=self.grads)
                    self.loss_acc[2] += self._part_loss[2]
                else:
                    self.grads.add_(self._part_grads, alpha=wgt)
                self.loss_acc[:2] += wgt * self._part_loss[:2]
                self.loss_acc[3] = torch.maximum(self.loss_acc[3], self._part_loss[3])

    def _plan_512(self, B, limit):
        """[(streams, run as)] for a batch of B streams at width 512.  The second-generation scans take every multiple of 512
        from 1024 to 3072 streams (32 row groups x 2 .. 6 row blocks of 16, lstm_scan2.hip) and are faster per stream than the
        first generation at ANY count above 512 (tools/probe_shape_sweep.py, and timed against the alternatives by
        test_stream_plan_is_the_fastest): so a batch is rounded up to whole blocks of 512 streams with dummy streams, and what no
        single launch sequence addresses (`limit`: 32-bit offsets into a layer's gate rows, T * B * 4W bf16) or holds (3072) is
        cut into groups of the largest count (3584 -> 2560 + 1024, 4096 -> 3072 + 1024, 6144 -> 2 x 3072).  No table of measured
        times: nothing here has to be re-measured when a kernel changes."""
        if self.plan_override is not None:      # (tests time the rule's choice against other plans: [(streams, run as), ...])
            if sum(n for n, _run in self.plan_override) == B:
                return list(self.plan_override)
            return [(B, dict(self.plan_override).get(B, B))]
        unit = 512
        max_units = min(6, limit // unit)
        if not self.pad_streams or B <= unit or max_units < 2:
            if B <= limit:
                return [(B, B)]
            # (windows so long that fewer than 1024 streams fit a launch sequence: groups of the largest size the kernels address)
            chunk = limit // unit * unit if limit >= unit else max(16, limit // 16 * 16)
            return [(min(chunk, B - b0), min(chunk, B - b0)) for b0 in range(0, B, chunk)]
        # groups of the largest count first (the more row blocks a workgroup serves per step, the cheaper a stream: 3072 + 1024
        # beats 2 x 2048), the rest in one group -- or two, if it would otherwise be a single block of 512 (first generation)
        units = -(-B // unit)
        sizes = []
        while units > max_units + 1:
            sizes.append(max_units)
            units -= max_units
        if units > max_units:
            sizes += [max_units - 1, 2]
        elif units > 0:
            sizes.append(units)
        plan, left = [], B
        for u in sizes:
            run = u * unit
            n = min(run, left)
            plan.append((n, run))
            left -= n
        return [(n, run) for n, run in plan if n > 0]

    def _padded_streams(self, n, T):
        """the stream count a group of n streams is run at: width 512 see _plan_512; width 1024: the eight-wave scans give each of
        their 8 row groups whole blocks of 16 streams, a step costs ~6.3 ms per started 128 streams (300 streams 21.6 ms, 384: 19.2;
        1000: 59.6, 1024: 50.0) -- rounded up to a multiple of 128; else n"""
        if T < 3 or self.max_streams_per_launch:
            return n
        if self.pwidth == 1024:
            up = -(-n // 128) * 128
            return up if (self.pad_streams and n > 128 and up <= 0xfffffff0 // (T * 4 * self.pwidth * 2)) else n
        if self.pwidth != 512:
            return n
        plan = self._plan_512(n, 0xfffffff0 // (T * 4 * self.pwidth * 2))
        return plan[0][1] if len(plan) == 1 else n

    def _rating_groups(self, B):
        """[(first, end)] stream ranges of a rating window (split precision): the persistent split-precision scans keep a
        workgroup's weights in registers, one workgroup per CU, at most four 16-stream row blocks each -- 256 streams at depth 2 /
        width 512, 128 at depth 3 or 4, 512 at width 256 (lstm_scan.hip, kl_launch_scan_fwd_split) --; more streams in one call
        fell through to the launch-per-step kernels (1024 x 256 characters: 62.6 ms against 4 x 4.3).  Streams are independent:
        a larger batch runs as groups of that many."""
        W, L = self.pwidth, self.depth
        cus = min(256, self.torch.cuda.get_device_properties(self.device).multi_processor_count)
        if W in (64, 128, 256, 512) and L <= 4:
            cap = 64 * (cus // (L * (W // 16)))
        elif W == 1024:
            cap = 64 * (cus // (W // 16))
        else:
            cap = 0
        if cap <= 0 or B <= cap:
            return [(0, B)]
        return [(b0, min(B, b0 + cap)) for b0 in range(0, B, cap)]

    # Stream counts per (physical) width that run on the fastest kernels, and the count from which a batch that is none of them
    # is regrouped: width 512 -- what the second-generation scans take (32 row groups x 2..6 row blocks of 16, lstm_scan2.hip);
    # width 1024 -- the eight-wave scans serve up to 1024 streams (lstm_scan_w32.hip), beyond them the launch-per-step kernels
    # (2048 streams: 1.4 M chars/s against 5.2 M).  (Width 128 runs any count in one piece since round 4: lstm_scan_w128.hip.)
    FAST_STREAMS = {512: ((3072, 2048, 1536, 1024), 1025), 1024: ((1024, 512), 1025)}

    def _stream_groups(self, B, T):
        """[(first, end)] stream ranges of one training batch.  One range, unless (a) the batch is beyond what a launch
        sequence addresses (32-bit offsets into a layer's gate rows, T * B * 4W bf16) -- it would fall through to the
        launch-per-step kernels --, or (b) it is none of the stream counts the fastest kernels of its width take and large
        enough (FAST_STREAMS): then the largest such counts are peeled off as long as 512 streams or more remain
        (width 512: 2560 -> 2048 + 512, 3584 -> 3072 + 512, 4096 -> 3072 + 1024) and only the rest runs on the slower path."""
        limit = self.max_streams_per_launch or (0xfffffff0 // (T * 4 * self.pwidth * 2))
        if self.pwidth == 512 and T >= 3 and not self.max_streams_per_launch:
            parts, b0 = [], 0
            for n, _run in self._plan_512(B, limit):
                parts.append((b0, b0 + n))
                b0 += n
            return parts
        fast, regroup_from = self.FAST_STREAMS.get(self.pwidth, ((), 0))
        fast = [f for f in fast if f <= limit] if (T >= 3 and not self.max_streams_per_launch) else []
        parts, b0 = [], 0
        while B - b0 > 0:
            rem = B - b0
            take = rem
            if fast and rem not in fast and (rem >= regroup_from or b0 > 0 or rem > limit):
                f = next((f for f in fast if f <= rem and (rem - f == 0 or rem - f >= 512)), None)
                if f is not None:
                    take = f
            if take > limit:      # (no fast count applies: groups of the largest size the kernels address)
                take = limit // 512 * 512 if limit >= 512 else max(16, limit // 16 * 16)
            parts.append((b0, b0 + take))
            b0 += take
        return parts

    def adam_step(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7, clip=1.0, grad_scale=1.0):
        """Keras-2.3 Adam(clipvalue=1.0) (rating.py:178).  grad_scale: the gradients are read as grads * grad_scale
        (1 / world size behind a summing all-reduce, distributed.GradSync.reduce)."""
        self.adam_t += 1
        with self._launch():
            hipabi.check(self.lib.kl_adam_step_scaled(self.handle, _ptr(self.grads), float(grad_scale), _ptr(self.adam_m),
                                                      _ptr(self.adam_v), self.adam_t, lr, b1, b2, eps, clip, self._stream()),
                         "kl_adam_step_scaled")

    def read_loss(self, reset=True):
        """(CE mean, accuracy, regulariser) accumulated since the last reset."""
        v = self.loss_acc.detach().cpu().numpy().copy()
        if reset:
            self.loss_acc.zero_()
        if v[3] != 0:
            raise hipabi.KlError("a persistent-scan hand-off timed out on the GPU (results of this window are invalid)")
        return float(v[0]), float(v[1]), float(v[2])

    # ------------------------------------------------------------------ incremental
    def ensure_pool(self, n_slots):
        torch = self.torch
        if self.pool is None or self.pool.shape[0] < n_slots:
            new = torch.zeros((n_slots, 2 * self.depth, self.pwidth), dtype=torch.float32, device=self.device)
            if self.pool is not None:
                new[:self.pool.shape[0]] = self.pool
            self.pool = new
        return self.pool

    def _i32(self, a):
        """device int32 view of `a` without a copy when it already is one"""
        torch = self.torch
        if isinstance(a, torch.Tensor) and a.dtype == torch.int32 and a.is_cuda and a.is_contiguous():
            return a
        return self._dev_i32(a)

    def step_slots(self, idx, ctx, slot_in, slot_out):
        """One LSTM step for n hypotheses whose states live in pool slots
        (device-resident variant of rating.py:578-639).  Returns probs tensor [n,V].

        A beam search calls this once per character, so the host side is kept short: the launches go
        onto the caller's current stream directly (every other engine call leaves that stream ordered
        after the engine stream, see _launch), nothing is copied when the arguments are device int32
        tensors, and the workspace size per n is cached."""
        torch = self.torch
        idx_d = self._i32(idx).reshape(-1)
        n = idx_d.numel()
        ctx_d = self._i32(ctx).reshape(n, -1) if self.n_ctx else None
        si = self._i32(slot_in).reshape(-1)
        so = self._i32(slot_out).reshape(-1)
        probs = torch.empty((n, self.voc_size), dtype=torch.float32, device=self.device)
        nws = self._step_ws_bytes.get(n)
        if nws is None:
            nws = self._step_ws_bytes[n] = int(self.lib.kl_step_workspace_bytes(self.handle, n))
        if self._step_ws is None or self._step_ws.numel() < nws:
            self._step_ws = torch.empty(nws, dtype=torch.uint8, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        hipabi.check(self.lib.kl_step_batch(self.handle, n, _ptr(idx_d), _ptr(ctx_d), _ptr(self.pool), _ptr(si),
                                            _ptr(so), _ptr(probs), _ptr(self._step_ws), self._step_ws.numel(),
                                            stream), "kl_step_batch")
        return probs

    def step_slots_heads(self, idx, ctx, slot_in, slot_out, k):
        """step_slots plus the first k state vectors of the new states, both on the HOST after ONE synchronising copy:
        (probs [n][V], heads [n][k][W]) as numpy arrays.  A beam search with history clustering (rating.py:887-916) needs
        both after every character; fetched one after the other they cost two round trips to an otherwise idle GPU."""
        torch = self.torch
        probs = self.step_slots(idx, ctx, slot_in, slot_out)
        so = self._i32(slot_out).reshape(-1)
        n = probs.shape[0]
        heads = self.pool.index_select(0, so.long())[:, :k, :self.width].reshape(n, -1)
        both = torch.cat([probs, heads], dim=1).cpu().numpy()
        return both[:, :self.voc_size], both[:, self.voc_size:].reshape(n, k, self.width)

    # ---- the step as a beam search issues it (kl_step_batch_host): host indices in, host results out, no stream synchronisation
    def _host_io(self, n, head_k):
        """page-locked, device-visible host buffers (kl_host_alloc) for up to n hypotheses: probabilities, head vectors,
        the arrival word; kept and grown as needed"""
        io = getattr(self, "_hio", None)
        if io is not None and io["n"] >= n and io["head_k"] >= head_k:
            return io
        if io is not None:
            self._host_io_free()
        n_cap = max(256, 1 << (n - 1).bit_length())
        sizes = {"probs": n_cap * self.voc_size * 4, "heads": max(1, n_cap * head_k * self.pwidth) * 4, "done": 64}
        io = {"n": n_cap, "head_k": head_k, "ticket": 0, "ptr": {}, "np": {}}
        for name, nbytes in sizes.items():
            ptr = self.lib.kl_host_alloc(nbytes)
            if not ptr:
                raise hipabi.KlError("kl_host_alloc(%d) failed" % nbytes)
            io["ptr"][name] = ptr
            ctype = C.c_uint32 if name == "done" else C.c_float
            io["np"][name] = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(nbytes // 4,))
        self._hio = io
        return io

    def _host_io_free(self):
        io = getattr(self, "_hio", None)
        if io is not None:
            self._hio = None
            self.torch.cuda.synchronize(self.device)      # (nothing may still be writing into the buffers)
            for ptr in io["ptr"].values():
                self.lib.kl_host_free(ptr)

    def step_host(self, idx, ctx, slot_in, slot_out, target=None, head_k=0, timeout=20.0):
        """One LSTM step for n hypotheses with HOST index arrays, results on the HOST (numpy): `Rater.predict`'s arithmetic
        (rating.py:578-639) at the latency a beam search sees.  Returns (probs, heads): probs [n][V], or [n] -- the
        probability of character target[i] for every row -- when `target` is given; heads [n][head_k][W] (the first head_k
        state vectors of the new states) or None.  The returned arrays are the caller's (copies of the delivery buffers)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1)
        n = idx.shape[0]
        ctx = np.ascontiguousarray(ctx, dtype=np.int32).reshape(n, -1) if self.n_ctx else None
        si = np.ascontiguousarray(slot_in, dtype=np.int32).reshape(-1)
        so = np.ascontiguousarray(slot_out, dtype=np.int32).reshape(-1)
        tg = np.ascontiguousarray(target, dtype=np.int32).reshape(-1) if target is not None else None
        io = self._host_io(n, head_k)
        nws = self._step_ws_bytes.get(("host", n))
        if nws is None:
            nws = self._step_ws_bytes[("host", n)] = int(self.lib.kl_step_host_workspace_bytes(self.handle, n))
        if self._step_host_ws is None or self._step_host_ws.numel() < nws:
            self._step_host_ws = self.torch.empty(nws, dtype=self.torch.uint8, device=self.device)
        io["ticket"] = ticket = (io["ticket"] % 0x7fffffff) + 1
        stream = C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)
        hipabi.check(self.lib.kl_step_batch_host(
            self.handle, n, idx.ctypes.data, ctx.ctypes.data if ctx is not None else None, si.ctypes.data, so.ctypes.data,
            tg.ctypes.data if tg is not None else None, _ptr(self.pool), int(head_k), io["ptr"]["probs"], io["ptr"]["heads"],
            io["ptr"]["done"], ticket, _ptr(self._step_host_ws), self._step_host_ws.numel(), stream), "kl_step_batch_host")
        hipabi.check(self.lib.kl_step_wait(io["ptr"]["done"], ticket, float(timeout)), "kl_step_wait")
        probs = io["np"]["probs"][:n].copy() if tg is not None else io["np"]["probs"][:n * self.voc_size].reshape(n, -1).copy()
        heads = None
        if head_k:
            heads = io["np"]["heads"][:n * head_k * self.pwidth].reshape(n, head_k, self.pwidth)[:, :, :self.width].copy()
        return probs, heads

    def to_device_i32(self, a):
        """one host-to-device transfer of an int32 array (rows stay contiguous views)"""
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device, non_blocking=True)

    def pool_heads(self, slots, k):
        """first k state vectors of the given slots, [n][k][W] on the host"""
        idx = self.torch.as_tensor(np.asarray(slots), device=self.device, dtype=self.torch.long)
        return self.pool[idx, :k, :self.width].cpu().numpy()

    def state_dist2(self, a, b, k):
        torch = self.torch
        a_d, b_d = self._dev_i32(a).reshape(-1), self._dev_i32(b).reshape(-1)
        out = torch.empty(a_d.numel(), dtype=torch.float32, device=self.device)
        with self._launch():
            hipabi.check(self.lib.kl_state_dist2(self.handle, a_d.numel(), _ptr(self.pool), _ptr(a_d), _ptr(b_d), int(k),
                                                 _ptr(out), self._stream()), "kl_state_dist2")
        return out

    # ------------------------------------------------------------------ pool access (host <-> HBM)
    def pool_zero(self, slots):
        with self._launch():
            self.pool[self.torch.as_tensor(list(slots), device=self.device, dtype=self.torch.long)] = 0

    def pool_read(self, slots):
        """states of the given slots as a numpy array [n][2L][W]"""
        with self._launch():
            out = self.pool[self.torch.as_tensor(list(slots), device=self.device, dtype=self.torch.long)][:, :, :self.width]
        return out.cpu().numpy()

    def pool_write(self, slots, values):
        with self._launch():
            v = self.torch.from_numpy(np.ascontiguousarray(values, dtype=np.float32)).to(self.device)
            if self.padded and v.shape[-1] == self.width:
                v = self.torch.nn.functional.pad(v, (0, self.pwidth - self.width))
            self.pool[self.torch.as_tensor(list(slots), device=self.device, dtype=self.torch.long)] = v
