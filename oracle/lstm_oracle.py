"""CPU oracle for the ocrd_keraslm Rater hot path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the arithmetic the reference delegates to
Keras 2.3 / TensorFlow 1.15 (neither is vendored in the reference nor installed
here).  It is the checker for the HIP path; nothing in `ocrd_keraslm_amd/`
imports it.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import this module.

PARITY PINNING STATUS
  * numerics (LSTM / softmax / CE / Adam): the reference holds no golden vectors
    at the Keras boundary and Keras/TF cannot be run here => **parity unpinned**
    against the reference itself.  The restatement is cross-checked against an
    independent implementation (torch.nn.LSTM + autograd on CPU,
    tests/test_oracle_vs_torch.py) and follows the call sites cited below.
  * host logic (windowing, beam search, Node): pinned by golden fixtures
    generated from the reference's own code (tests/golden/make_golden.py).

Reference call sites restated (paths relative to /root/reference/ocrd_keraslm/lib):
  rating.py:103-125   Embedding(voc,width) + Embedding(200,10) + Concatenate
  rating.py:126-152   depth x LSTM(width), gate order i,f,c,o, sigmoid recurrent
                      activation; Dropout(0.1, noise_shape time-constant) after
                      every layer with index > 0
  rating.py:155-168   output = softmax(h . E^T)   (tied embedding, no bias)
  rating.py:178       loss categorical_crossentropy, Adam(clipvalue=1.0), accuracy
  rating.py:187-246   embedding regularisers (train phase only)
  rating.py:578-639   incremental single step with explicit states
  callbacks.py:36-69  state reset points (host side; see tests)

Keras/TF semantics restated:
  LSTM cell       z = x.K + h.U + b ; i=sig(z0) f=sig(z1) g=tanh(z2) o=sig(z3)
                  c' = f*c + i*g ; h' = o*tanh(c')
  crossentropy    p <- p/sum(p); p <- clip(p,1e-7,1-1e-7); l = -sum(y log p);
                  mean over ALL B*T positions (all-zero target rows count)
  Adam (2.3.1)    g <- clip(g,-1,1); t=it+1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
                  m=b1 m+(1-b1) g; v=b2 v+(1-b2) g^2; p -= lr_t*m/(sqrt(v)+1e-7)
"""
from __future__ import annotations

import numpy as np

CTX_VOCAB = 200   # rating.py:111
CTX_DIM = 10      # rating.py:111
DROPOUT_RATE = 0.1  # rating.py:152


class ModelConfig:
    """Topology of one Rater network (rating.py:39-59, 61-179)."""

    def __init__(self, depth, width, voc_size, n_ctx=1):
        self.depth = int(depth)
        self.width = int(width)
        self.voc_size = int(voc_size)
        self.n_ctx = int(n_ctx)

    @property
    def in_dim0(self):
        return self.width + CTX_DIM * self.n_ctx

    def in_dim(self, layer):
        return self.in_dim0 if layer == 0 else self.width

    def param_shapes(self):
        """Ordered (name, shape) list = Keras weight order of the weighted layers
        char_embedding, context{n}_embedding, lstm_1..lstm_L."""
        W = self.width
        shapes = [("E", (self.voc_size, W))]
        for n in range(self.n_ctx):
            shapes.append(("Ctx%d" % n, (CTX_VOCAB, CTX_DIM)))
        for l in range(self.depth):
            shapes.append(("K%d" % l, (self.in_dim(l), 4 * W)))
            shapes.append(("U%d" % l, (W, 4 * W)))
            shapes.append(("b%d" % l, (4 * W,)))
        return shapes

    def param_count(self):
        return sum(int(np.prod(s)) for _, s in self.param_shapes())

    def flops_fwd_per_char(self):
        """F_fwd = sum_l 2 (D_l+W) 4W + 2 W V."""
        W = self.width
        return sum(2 * (self.in_dim(l) + W) * 4 * W for l in range(self.depth)) + 2 * W * self.voc_size


def _orthogonal(rng, rows, cols):
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return q[:rows, :cols]


def init_weights(cfg, seed=1, emb_std=0.001, dtype=np.float32):
    """Keras default initialisers as configured at rating.py:104-114 + LSTM
    defaults (glorot_uniform kernel, orthogonal recurrent, zero bias with unit
    forget bias).  emb_std=0.5 gives the 'trained-like' weights of SURVEY 8(d)."""
    rng = np.random.default_rng(seed)
    W = cfg.width
    w = {}
    w["E"] = (rng.standard_normal((cfg.voc_size, W)) * emb_std).astype(dtype)
    for n in range(cfg.n_ctx):
        w["Ctx%d" % n] = (rng.standard_normal((CTX_VOCAB, CTX_DIM)) * emb_std).astype(dtype)
    for l in range(cfg.depth):
        D = cfg.in_dim(l)
        lim = np.sqrt(6.0 / (D + 4 * W))
        w["K%d" % l] = rng.uniform(-lim, lim, (D, 4 * W)).astype(dtype)
        w["U%d" % l] = _orthogonal(rng, W, 4 * W).astype(dtype)
        b = np.zeros(4 * W, dtype=dtype)
        b[W:2 * W] = 1.0
        w["b%d" % l] = b
    return w


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def softmax(logits):
    m = logits.max(axis=-1, keepdims=True)
    e = np.exp(logits - m)
    return e / e.sum(axis=-1, keepdims=True)


def bf16_round(x):
    """Round-to-nearest-even f32 -> bf16 -> f32 (used to model the HIP bf16 path)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def embed(cfg, w, idx, ctx):
    """F1: X0[..., :] = [E[idx] || Ctx0[ctx0] || ...]  (rating.py:103-125).
    idx: int [...], ctx: int [..., n_ctx]."""
    parts = [w["E"][idx]]
    for n in range(cfg.n_ctx):
        parts.append(w["Ctx%d" % n][ctx[..., n]])
    return np.concatenate(parts, axis=-1)


def zero_states(cfg, batch, dtype=np.float32):
    return [np.zeros((batch, cfg.width), dtype=dtype) for _ in range(2 * cfg.depth)]


def lstm_cell(x, h, c, K, U, b, W):
    z = x @ K + h @ U + b
    i = sigmoid(z[:, 0:W])
    f = sigmoid(z[:, W:2 * W])
    g = np.tanh(z[:, 2 * W:3 * W])
    o = sigmoid(z[:, 3 * W:4 * W])
    c2 = f * c + i * g
    h2 = o * np.tanh(c2)
    return h2, c2, (i, f, g, o)


def step_batch(cfg, w, idx, ctx, states):
    """S1 (rating.py:578-639): one time step for n rows with explicit state.
    idx [n], ctx [n, n_ctx], states = [h1,c1,...,hL,cL] each [n, W].
    Returns probs [n, V], new states (same order)."""
    W = cfg.width
    x = embed(cfg, w, idx, ctx)
    out = []
    for l in range(cfg.depth):
        h, c, _ = lstm_cell(x, states[2 * l], states[2 * l + 1],
                            w["K%d" % l], w["U%d" % l], w["b%d" % l], W)
        out += [h, c]
        x = h
    probs = softmax(x @ w["E"].T)
    return probs, out


def forward_window(cfg, w, idx, ctx, states, dropout_masks=None, keep_cache=False):
    """F1-F5 over one window.  idx [B,T] int, ctx [B,T,n_ctx] int,
    states [h1,c1,..] each [B,W] (carried-in, rating.py:127-128 stateful).
    dropout_masks: None (inference) or list indexed by layer of [B,W] keep-masks
    already scaled by 1/0.9 (None for layer 0) -- time-constant (rating.py:146-152).
    Returns probs [B,T,V], final states, cache (for backward)."""
    W = cfg.width
    B, T = idx.shape
    x = embed(cfg, w, idx, ctx)  # [B,T,D0]
    cache = {"x": [], "gates": [], "c": [], "h": [], "hpre": [], "states_in": [s.copy() for s in states]}
    new_states = []
    for l in range(cfg.depth):
        K, U, b = w["K%d" % l], w["U%d" % l], w["b%d" % l]
        h, c = states[2 * l], states[2 * l + 1]
        hs = np.empty((B, T, W), dtype=x.dtype)
        if keep_cache:
            gs = np.empty((B, T, 4, W), dtype=x.dtype)
            cs = np.empty((B, T, W), dtype=x.dtype)
        for t in range(T):
            h, c, (i, f, g, o) = lstm_cell(x[:, t], h, c, K, U, b, W)
            hs[:, t] = h
            if keep_cache:
                gs[:, t, 0], gs[:, t, 1], gs[:, t, 2], gs[:, t, 3] = i, f, g, o
                cs[:, t] = c
        new_states += [h, c]
        if keep_cache:
            cache["x"].append(x)
            cache["gates"].append(gs)
            cache["c"].append(cs)
            cache["hpre"].append(hs)
        if dropout_masks is not None and l > 0 and dropout_masks[l] is not None:
            x = hs * dropout_masks[l][:, None, :]
        else:
            x = hs
        if keep_cache:
            cache["h"].append(x)
    logits = x @ w["E"].T
    probs = softmax(logits)
    return probs, new_states, cache


def crossentropy(probs, tgt):
    """F6 (rating.py:178; Keras TF backend categorical_crossentropy + accuracy).
    probs [B,T,V]; tgt int [B,T] with -1 meaning an all-zero one-hot row
    (the padded tail of the last window, rating.py:1096-1102, 1123-1157).
    Returns (mean loss over all B*T positions, accuracy, per-position loss)."""
    B, T, V = probs.shape
    p = probs / probs.sum(axis=-1, keepdims=True)
    p = np.clip(p, 1e-7, 1 - 1e-7)
    valid = tgt >= 0
    tsafe = np.where(valid, tgt, 0)
    pt = np.take_along_axis(p, tsafe[..., None], axis=-1)[..., 0]
    l = np.where(valid, -np.log(pt), 0.0)
    # accuracy: argmax of an all-zero target row is 0
    acc = (np.argmax(probs, axis=-1) == tsafe).mean()
    return l.mean(), acc, l


def regularisers(cfg, w):
    """F7 (rating.py:187-246), value in the training phase."""
    E = w["E"].astype(np.float64)
    total = 0.0
    if E.shape[0] > 0:
        m = E[1:].mean(axis=0)
        total += ((E[0] - m) ** 2).sum()
        n = (E ** 2).sum(axis=1)
        total += 0.01 * ((1 - n) ** 2).sum()
    for k in range(cfg.n_ctx):
        C = w["Ctx%d" % k].astype(np.float64)
        n = (C ** 2).sum(axis=1)
        total += 0.02 * ((1 - n) ** 2).sum()
        total += 0.2 * (C[1:-1].sum(axis=0) * C[2:].sum(axis=0)).sum()
        m = C[1:].mean(axis=0)
        total += 2.0 * ((C[0][None, :] - n[1:, None] * m[None, :]) ** 2).sum()
    return total


def regulariser_grads(cfg, w):
    """Gradients of `regularisers` honouring the stop_gradient placements
    (rating.py:205, 212-213, 236)."""
    g = {}
    E = w["E"].astype(np.float64)
    gE = np.zeros_like(E)
    if E.shape[0] > 0:
        m = E[1:].mean(axis=0)
        gE[0] += 2 * (E[0] - m)
        n = (E ** 2).sum(axis=1)
        gE += (-0.04 * (1 - n))[:, None] * E
    g["E"] = gE
    for k in range(cfg.n_ctx):
        C = w["Ctx%d" % k].astype(np.float64)
        gC = np.zeros_like(C)
        n = (C ** 2).sum(axis=1)
        gC += (-0.08 * (1 - n))[:, None] * C
        gC[2:] += 0.2 * C[1:-1].sum(axis=0)[None, :]
        m = C[1:].mean(axis=0)
        gC[0] += 4.0 * (C[0][None, :] - n[1:, None] * m[None, :]).sum(axis=0)
        g["Ctx%d" % k] = gC
    return g


def backward_window(cfg, w, idx, ctx, tgt, probs, cache, dropout_masks=None, with_regularisers=True):
    """B1-B7: gradient of (mean CE [+ regularisers]) w.r.t. all weights.
    Truncated BPTT: no gradient into the carried-in states."""
    W = cfg.width
    B, T, V = probs.shape
    dt = probs.dtype
    grads = {name: np.zeros(shape, dtype=np.float64) for name, shape in cfg.param_shapes()}
    valid = (tgt >= 0)
    tsafe = np.where(valid, tgt, 0)
    # dlogits = ((sum y) p - y) / (B T); zero where the clip is active
    pt = np.take_along_axis(probs, tsafe[..., None], axis=-1)[..., 0]
    active = valid & (pt >= 1e-7) & (pt <= 1 - 1e-7)
    dlog = probs * active[..., None]
    onehot = np.zeros_like(probs)
    np.put_along_axis(onehot, tsafe[..., None], active[..., None].astype(dt), axis=-1)
    dlog = (dlog - onehot) / (B * T)
    hL = cache["h"][-1]
    grads["E"] += dlog.reshape(-1, V).T.astype(np.float64) @ hL.reshape(-1, W).astype(np.float64)
    dx = dlog @ w["E"]  # [B,T,W]
    for l in reversed(range(cfg.depth)):
        if dropout_masks is not None and l > 0 and dropout_masks[l] is not None:
            dx = dx * dropout_masks[l][:, None, :]
        K, U = w["K%d" % l], w["U%d" % l]
        gs, cs, xs, hs = cache["gates"][l], cache["c"][l], cache["x"][l], cache["hpre"][l]
        h0, c0 = cache["states_in"][2 * l], cache["states_in"][2 * l + 1]
        dz = np.empty((B, T, 4 * W), dtype=dt)
        dh_rec = np.zeros((B, W), dtype=dt)
        dc = np.zeros((B, W), dtype=dt)
        for t in reversed(range(T)):
            i, f, g, o = gs[:, t, 0], gs[:, t, 1], gs[:, t, 2], gs[:, t, 3]
            c = cs[:, t]
            cprev = cs[:, t - 1] if t > 0 else c0
            tc = np.tanh(c)
            dh = dx[:, t] + dh_rec
            do = dh * tc
            dc = dc + dh * o * (1 - tc * tc)
            di = dc * g
            dg = dc * i
            df = dc * cprev
            dz[:, t, 0:W] = di * i * (1 - i)
            dz[:, t, W:2 * W] = df * f * (1 - f)
            dz[:, t, 2 * W:3 * W] = dg * (1 - g * g)
            dz[:, t, 3 * W:4 * W] = do * o * (1 - o)
            dh_rec = dz[:, t] @ U.T
            dc = dc * f
        hprev = np.concatenate([h0[:, None, :], hs[:, :-1]], axis=1)
        dz2 = dz.reshape(-1, 4 * W).astype(np.float64)
        grads["U%d" % l] += hprev.reshape(-1, W).astype(np.float64).T @ dz2
        grads["K%d" % l] += xs.reshape(-1, xs.shape[-1]).astype(np.float64).T @ dz2
        grads["b%d" % l] += dz2.sum(axis=0)
        dx = dz @ K.T
    # scatter into embeddings
    np.add.at(grads["E"], idx.reshape(-1), dx[..., :W].reshape(-1, W).astype(np.float64))
    for n in range(cfg.n_ctx):
        sl = slice(W + n * CTX_DIM, W + (n + 1) * CTX_DIM)
        np.add.at(grads["Ctx%d" % n], ctx[..., n].reshape(-1), dx[..., sl].reshape(-1, CTX_DIM).astype(np.float64))
    if with_regularisers:
        rg = regulariser_grads(cfg, w)
        for k, v in rg.items():
            grads[k] += v
    return grads


class Adam:
    """O1: Keras 2.3.1 Adam(clipvalue=1.0) (rating.py:178)."""

    def __init__(self, cfg, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7, clipvalue=1.0, dtype=np.float32):
        self.lr, self.b1, self.b2, self.eps, self.clip = lr, b1, b2, eps, clipvalue
        self.t = 0
        self.m = {n: np.zeros(s, dtype=dtype) for n, s in cfg.param_shapes()}
        self.v = {n: np.zeros(s, dtype=dtype) for n, s in cfg.param_shapes()}

    def step(self, w, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for n in w:
            g = np.clip(grads[n], -self.clip, self.clip).astype(w[n].dtype)
            self.m[n] = self.b1 * self.m[n] + (1 - self.b1) * g
            self.v[n] = self.b2 * self.v[n] + (1 - self.b2) * g * g
            w[n] = (w[n] - lr_t * self.m[n] / (np.sqrt(self.v[n]) + self.eps)).astype(w[n].dtype)


def draw_dropout_masks(cfg, batch, rng, dtype=np.float32):
    """Inverted dropout keep-masks, one per stream and layer index > 0, constant
    over the window (rating.py:146-152)."""
    masks = [None]
    for l in range(1, cfg.depth):
        keep = rng.random((batch, cfg.width)) >= DROPOUT_RATE
        masks.append((keep / (1.0 - DROPOUT_RATE)).astype(dtype))
    return masks


def train_step(cfg, w, opt, idx, ctx, tgt, states, dropout_masks=None):
    """One fit_generator batch (rating.py:292-298): forward, loss (+regularisers),
    backward, clip+Adam.  Mutates w / opt; returns (loss, acc, new states)."""
    probs, new_states, cache = forward_window(cfg, w, idx, ctx, states, dropout_masks, keep_cache=True)
    ce, acc, _ = crossentropy(probs, tgt)
    loss = ce + regularisers(cfg, w)
    grads = backward_window(cfg, w, idx, ctx, tgt, probs, cache, dropout_masks)
    opt.step(w, grads)
    return float(loss), float(acc), new_states
