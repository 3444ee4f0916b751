"""Test double for `engine.HipLM` backed by the numpy oracle -- TEST INFRASTRUCTURE.

Lets the host logic of `Rater` (windowing, beam searches, training-loop control,
model I/O, data-parallel gradient averaging) be exercised without a GPU by the
`-m "not gpu"` suite.  The product never uses it: `Rater` defaults to HipLM and
HipLM raises without a GPU / without the HIP library."""
import numpy as np

from oracle import lstm_oracle as O


class OracleLM(object):
    def __init__(self, depth, width, voc_size, n_ctx=1, dtype=np.float64):
        self.depth, self.width, self.voc_size, self.n_ctx = depth, width, voc_size, n_ctx
        self.cfg = O.ModelConfig(depth, width, voc_size, n_ctx)
        self.dtype = dtype
        self.w = O.init_weights(self.cfg, seed=0, dtype=dtype)
        self.layout = []
        off = 0
        for name, shape in self.cfg.param_shapes():
            rows, cols = (1, shape[0]) if len(shape) == 1 else shape
            self.layout.append((name, off, rows, cols))
            off += rows * cols
        self.n_params = off
        self.precision = 3
        self.states = None
        self.pool = None
        self.grads = None
        self.opt = None
        self.loss = np.zeros(3)
        self._rng = np.random.default_rng(0)
        self.step_calls = []
        self.last_only = False

    def set_window_mode(self, last_only):
        self.last_only = bool(last_only)

    def _loss(self, probs, tgt):
        """(mean CE, accuracy) in the current window mode"""
        if not self.last_only:
            ce, acc, _ = O.crossentropy(probs, tgt)
            return ce, acc
        last = np.asarray(tgt)[:, -1]
        p = probs[:, -1]
        has = last >= 0
        pc = np.clip(p[np.arange(len(last)), np.where(has, last, 0)], 1e-7, 1 - 1e-7)
        return float(np.mean(np.where(has, -np.log(pc), 0.0))), float(np.mean(p.argmax(axis=-1) == np.where(has, last, 0)))

    # weights
    def get_weights(self):
        return {k: np.array(v, dtype=np.float32) for k, v in self.w.items()}

    def set_weights(self, weights, precision=None):
        self.w = {k: np.asarray(weights[k], dtype=self.dtype).reshape(self.w[k].shape) for k in self.w}
        if precision:
            self.precision = precision

    def init_weights(self, seed=None, emb_std=0.001):
        self.w = O.init_weights(self.cfg, seed=seed if seed is not None else 0, emb_std=emb_std, dtype=self.dtype)

    def prepare(self, precision):
        self.precision = precision

    # windows
    def reset_states(self, B=None, rows=None):
        if self.states is None or (B is not None and self.states[0].shape[0] != B):
            self.states = O.zero_states(self.cfg, B or 1, self.dtype)
        elif rows is None:
            self.states = O.zero_states(self.cfg, self.states[0].shape[0], self.dtype)
        else:
            for s in self.states:
                s[list(rows)] = 0

    def forward_window(self, idx, ctx, tgt=None, want_probs=True):
        idx = np.asarray(idx).astype(np.int64)
        ctx = np.asarray(ctx).astype(np.int64).reshape(idx.shape + (self.n_ctx,))
        if self.states is None or self.states[0].shape[0] != idx.shape[0]:
            self.reset_states(idx.shape[0])
        probs, self.states, _ = O.forward_window(self.cfg, self.w, idx, ctx, self.states)
        if tgt is not None:
            ce, acc = self._loss(probs, np.asarray(tgt))
            self.loss[0] += ce
            self.loss[1] += acc
        return probs if want_probs else None

    def get_grads(self):
        flat = self.grads.numpy()
        return {name: flat[off:off + rows * cols].reshape(self.w[name].shape).copy() for name, off, rows, cols in self.layout}

    def draw_dropout_masks(self, B):
        keep = self._rng.random((self.depth, B, self.width)) >= O.DROPOUT_RATE
        m = (keep / (1.0 - O.DROPOUT_RATE)).astype(np.float32)
        m[0] = 1.0
        return m

    def ensure_training_buffers(self):
        if self.opt is None:
            self.opt = O.Adam(self.cfg, dtype=self.dtype)

    def train_window(self, idx, ctx, tgt, masks=None):
        import torch
        self.ensure_training_buffers()
        idx = np.asarray(idx).astype(np.int64)
        ctx = np.asarray(ctx).astype(np.int64).reshape(idx.shape + (self.n_ctx,))
        tgt = np.asarray(tgt)
        if self.states is None or self.states[0].shape[0] != idx.shape[0]:
            self.reset_states(idx.shape[0])
        om = None
        if masks is not None:
            om = [None] + [np.asarray(masks[l], dtype=self.dtype) for l in range(1, self.depth)]
        probs, self.states, cache = O.forward_window(self.cfg, self.w, idx, ctx, self.states, om, keep_cache=True)
        ce, acc = self._loss(probs, tgt)
        self.loss += (ce, acc, O.regularisers(self.cfg, self.w))
        if self.last_only:       # mean over the B windows instead of the B*T positions
            tl = np.full(tgt.shape, -1)
            tl[:, -1] = tgt[:, -1]
            g_all = O.backward_window(self.cfg, self.w, idx, ctx, tl, probs, cache, om)
            g_ce = O.backward_window(self.cfg, self.w, idx, ctx, tl, probs, cache, om, with_regularisers=False)
            g = {k: idx.shape[1] * g_ce[k] + (g_all[k] - g_ce[k]) for k in g_all}
        else:
            g = O.backward_window(self.cfg, self.w, idx, ctx, tgt, probs, cache, om)
        flat = np.zeros(self.n_params, dtype=np.float32)
        for name, off, rows, cols in self.layout:
            flat[off:off + rows * cols] = g[name].reshape(-1)
        self.grads = torch.from_numpy(flat)     # a CPU tensor: what GradSync all-reduces over gloo

    def adam_step(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7, clip=1.0, grad_scale=1.0):
        flat = self.grads.numpy() * np.float32(grad_scale)
        g = {name: flat[off:off + rows * cols].reshape(self.w[name].shape) for name, off, rows, cols in self.layout}
        self.opt.step(self.w, g)

    def read_loss(self, reset=True):
        v = tuple(float(x) for x in self.loss)
        if reset:
            self.loss[:] = 0
        return v

    # incremental
    def ensure_pool(self, n_slots):
        if self.pool is None:
            self.pool = np.zeros((n_slots, 2 * self.depth, self.width), dtype=self.dtype)
        elif self.pool.shape[0] < n_slots:
            new = np.zeros((n_slots, 2 * self.depth, self.width), dtype=self.dtype)
            new[:self.pool.shape[0]] = self.pool
            self.pool = new
        return self.pool

    def pool_zero(self, slots):
        self.pool[list(slots)] = 0

    def pool_read(self, slots):
        return self.pool[list(slots)].copy()

    def pool_write(self, slots, values):
        self.pool[list(slots)] = np.asarray(values)

    def step_slots(self, idx, ctx, slot_in, slot_out):
        idx = np.asarray(idx).astype(np.int64).reshape(-1)
        n = len(idx)
        ctx = np.asarray(ctx).astype(np.int64).reshape(n, self.n_ctx)
        si, so = np.asarray(slot_in).reshape(-1), np.asarray(slot_out).reshape(-1)
        assert not set(si.tolist()) & set(so.tolist()), "slot_out must not alias slot_in"
        states = [self.pool[si, k] for k in range(2 * self.depth)]
        probs, new = O.step_batch(self.cfg, self.w, idx, ctx, states)
        for k in range(2 * self.depth):
            self.pool[so, k] = new[k]
        self.step_calls.append(n)
        return probs.astype(np.float32) if self.dtype == np.float32 else probs

    def pool_heads(self, slots, k):
        return self.pool[list(slots), :k].copy()

    def state_dist2(self, a, b, k):
        d = self.pool[list(a), k] - self.pool[list(b), k]
        return (d * d).sum(axis=-1)
