"""Pin the numpy oracle against an independent implementation: torch.nn.LSTM
(oneDNN/ATen CPU kernels) + autograd.  Keras weight layout -> torch:
weight_ih = K^T, weight_hh = U^T, bias_ih = b, bias_hh = 0; Keras gate order
i,f,c,o == torch gate order i,f,g,o."""
import numpy as np
import pytest
import torch

from oracle import lstm_oracle as O


def torch_model(cfg, w, dtype=torch.float64):
    p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in w.items()}
    return p


def torch_forward(cfg, p, idx, ctx, states, masks=None):
    W = cfg.width
    ti = torch.as_tensor(idx, dtype=torch.long)
    x = [p["E"][ti]]
    for n in range(cfg.n_ctx):
        x.append(p["Ctx%d" % n][torch.as_tensor(ctx[..., n], dtype=torch.long)])
    x = torch.cat(x, dim=-1)
    new_states = []
    for l in range(cfg.depth):
        w_ih = p["K%d" % l].t()
        w_hh = p["U%d" % l].t()
        h0 = torch.as_tensor(states[2 * l], dtype=x.dtype)[None]
        c0 = torch.as_tensor(states[2 * l + 1], dtype=x.dtype)[None]
        out, hn, cn = torch._VF.lstm(x, (h0, c0), [w_ih.contiguous(), w_hh.contiguous(), p["b%d" % l],
                                                   torch.zeros_like(p["b%d" % l])],
                                     True, 1, 0.0, False, False, True)
        new_states += [hn[0], cn[0]]
        if masks is not None and l > 0 and masks[l] is not None:
            out = out * torch.as_tensor(masks[l], dtype=x.dtype)[:, None, :]
        x = out
    logits = x @ p["E"].t()
    return torch.softmax(logits, dim=-1), new_states


def torch_loss(cfg, p, probs, tgt, train=True):
    B, T, V = probs.shape
    q = probs / probs.sum(-1, keepdim=True)
    q = torch.clamp(q, 1e-7, 1 - 1e-7)
    y = torch.zeros_like(probs)
    tt = torch.as_tensor(tgt, dtype=torch.long)
    valid = tt >= 0
    y[valid.nonzero(as_tuple=True) + (tt[valid],)] = 1.0
    ce = -(y * torch.log(q)).sum(-1).mean()
    if not train:
        return ce
    E = p["E"]
    reg = ((E[0] - E[1:].mean(0).detach()) ** 2).sum() + 0.01 * ((1 - (E ** 2).sum(1)) ** 2).sum()
    for n in range(cfg.n_ctx):
        C = p["Ctx%d" % n]
        norms = (C ** 2).sum(1)
        reg = reg + 0.02 * ((1 - norms) ** 2).sum()
        reg = reg + 0.2 * (C[1:-1].detach() @ C[2:].t()).sum()
        wg = (C[1:] * C[1:]).sum(1, keepdim=True).detach()
        mean = C[1:].mean(0).detach()
        reg = reg + 2 * ((C[0:1] - wg * mean) ** 2).sum()
    return ce + reg


def make_case(depth, width, voc, n_ctx, B, T, seed, pad_tail=0):
    cfg = O.ModelConfig(depth, width, voc, n_ctx)
    w = O.init_weights(cfg, seed=seed, emb_std=0.3, dtype=np.float64)
    rng = np.random.default_rng(seed + 100)
    idx = rng.integers(0, voc, (B, T))
    ctx = rng.integers(0, O.CTX_VOCAB, (B, 1, n_ctx)).repeat(T, axis=1)
    tgt = rng.integers(0, voc, (B, T))
    if pad_tail:
        idx[:, -pad_tail:] = 0
        ctx[:, -pad_tail:] = 0
        tgt[:, -pad_tail:] = -1
    states = [rng.standard_normal((B, width)) * 0.1 for _ in range(2 * depth)]
    return cfg, w, idx, ctx, tgt, states


@pytest.mark.parametrize("depth,width,voc,n_ctx,B,T", [(1, 8, 7, 1, 2, 5), (2, 16, 11, 1, 3, 9), (3, 12, 9, 2, 2, 6)])
def test_forward_matches_torch(depth, width, voc, n_ctx, B, T):
    cfg, w, idx, ctx, tgt, states = make_case(depth, width, voc, n_ctx, B, T, seed=3)
    probs, ns, _ = O.forward_window(cfg, w, idx, ctx, states)
    p = torch_model(cfg, w)
    tp, tns = torch_forward(cfg, p, idx, ctx, states)
    assert np.abs(probs - tp.detach().numpy()).max() < 1e-12
    for a, b in zip(ns, tns):
        assert np.abs(a - b.detach().numpy()).max() < 1e-12


def test_step_equals_window():
    cfg, w, idx, ctx, tgt, states = make_case(2, 16, 11, 1, 3, 9, seed=5)
    probs, ns, _ = O.forward_window(cfg, w, idx, ctx, states)
    st = [s.copy() for s in states]
    for t in range(idx.shape[1]):
        pr, st = O.step_batch(cfg, w, idx[:, t], ctx[:, t], st)
        assert np.abs(pr - probs[:, t]).max() < 1e-13
    for a, b in zip(ns, st):
        assert np.abs(a - b).max() < 1e-13


@pytest.mark.parametrize("depth,width,voc,n_ctx,B,T,pad", [(1, 8, 7, 1, 2, 5, 0), (2, 16, 11, 1, 3, 9, 4), (3, 12, 9, 2, 2, 6, 0)])
def test_gradients_match_autograd(depth, width, voc, n_ctx, B, T, pad):
    cfg, w, idx, ctx, tgt, states = make_case(depth, width, voc, n_ctx, B, T, seed=7, pad_tail=pad)
    rng = np.random.default_rng(11)
    masks = O.draw_dropout_masks(cfg, B, rng, dtype=np.float64)
    probs, ns, cache = O.forward_window(cfg, w, idx, ctx, states, masks, keep_cache=True)
    ce, acc, _ = O.crossentropy(probs, tgt)
    loss = ce + O.regularisers(cfg, w)
    grads = O.backward_window(cfg, w, idx, ctx, tgt, probs, cache, masks)
    p = torch_model(cfg, w)
    tp, _ = torch_forward(cfg, p, idx, ctx, states, masks)
    tl = torch_loss(cfg, p, tp, tgt)
    assert abs(float(tl) - loss) < 1e-10
    tl.backward()
    for k in w:
        tg = p[k].grad.numpy()
        assert np.abs(grads[k] - tg).max() < 1e-10 * max(1.0, np.abs(tg).max()), k


def test_adam_matches_formula():
    cfg = O.ModelConfig(1, 4, 5, 1)
    w = O.init_weights(cfg, seed=1, dtype=np.float64)
    w0 = {k: v.copy() for k, v in w.items()}
    opt = O.Adam(cfg, dtype=np.float64)
    g = {k: np.full_like(v, 3.0) for k, v in w.items()}   # clipped to 1
    opt.step(w, g)
    # t=1: lr_t = lr*sqrt(1-b2)/(1-b1); m = 0.1, v = 0.001 -> update = lr_t*0.1/(sqrt(.001)+1e-7)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    upd = lr_t * 0.1 / (np.sqrt(0.001) + 1e-7)
    for k in w:
        assert np.allclose(w0[k] - w[k], upd, rtol=1e-12)
    # torch.optim.Adam with eps scaled is NOT identical (eps inside bias correction) -- formula check only


def test_crossentropy_counts_padding():
    probs = np.full((1, 4, 5), 0.2)
    tgt = np.array([[1, 2, -1, -1]])
    ce, acc, l = O.crossentropy(probs, tgt)
    assert np.isclose(ce, 2 * -np.log(0.2) / 4)
    # argmax of uniform probs = 0; padded targets have argmax 0 -> counted as hits
    assert np.isclose(acc, 0.5)
