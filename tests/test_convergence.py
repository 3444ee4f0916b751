"""Training-convergence parity.

The reference pins its training path only by perplexity (test/test_wrapper.py:49-50, 101-102: a model trained by
`keraslm-rate train` must rate the test text below a perplexity bound; Makefile:82-88 trains, then tests).  The same kind of
pin, HIP engine against the f32 restatement: depth 2 / width 128 / length 64, 8 stateful streams of a seeded order-1 Markov
text (the generator of test_rater_plumbing.synth_files), 300 steps of forward + backward + clip + Adam (rating.py:178,
292-298) from the same initial weights, with the same batches, the same dropout masks and carried states -- the bf16 path
with its fast gates, bf16 P / Cb / dH hand-offs on one side, f32 numpy on the other.  Held-out windows are rated from zero
state every 50 steps: the two validation-loss curves must stay within 2 % of each other, the final perplexities within 3 %,
and the model must actually have learnt the chain (validation loss well below that of the untrained model)."""
import json
import os
import time

import numpy as np
import pytest

from oracle import lstm_oracle as O

pytestmark = pytest.mark.gpu

CHARS = "abcdefghijklmnopqrstuvwxyz ABCDEFG.,;!?\n-"


def markov_ids(rng, trans, n_streams, length):
    """[n_streams][length] character ids 1 .. len(CHARS) (0 = unmapped, never produced) of the seeded chain"""
    out = np.empty((n_streams, length), dtype=np.int64)
    for b in range(n_streams):
        s = int(rng.integers(len(CHARS)))
        for i in range(length):
            out[b, i] = s + 1
            s = int(rng.choice(len(CHARS), p=trans[s]))
    return out


@pytest.mark.timeout(600)
def test_bf16_training_follows_the_f32_restatement():
    from ocrd_keraslm_amd.lib import hipabi
    from ocrd_keraslm_amd.lib.engine import HipLM
    depth, width, voc, n_ctx, B, T, steps, every = 2, 128, len(CHARS) + 1, 1, 8, 64, 300, 50
    rng = np.random.default_rng(1)
    trans = rng.dirichlet(np.full(len(CHARS), 0.05), size=len(CHARS))
    text = markov_ids(rng, trans, B, steps * T + 1)
    held = markov_ids(rng, trans, B, 4 * T + 1)
    ctx = np.full((B, T, n_ctx), 84, dtype=np.int64)                  # (one context value: "year" 1784, clamped as the Rater would)
    cfg = O.ModelConfig(depth, width, voc, n_ctx)
    w0 = O.init_weights(cfg, seed=11, emb_std=0.05, dtype=np.float32)
    lm = HipLM(depth, width, voc, n_ctx)
    lm.set_weights(w0, hipabi.KL_PREC_BF16)
    lm.reset_states(B)
    lm.ensure_training_buffers()
    wo = {k: v.copy() for k, v in w0.items()}
    opt = O.Adam(cfg, dtype=np.float32)
    st = O.zero_states(cfg, B, np.float32)

    def validate():
        """mean CE of the held-out windows from zero state: (HIP in its training precision, f32 restatement)"""
        carried = lm.get_states()
        ref, got = [], []
        for k in range(4):
            vi, vt = held[:, k * T:(k + 1) * T], held[:, k * T + 1:(k + 1) * T + 1]
            p, _, _ = O.forward_window(cfg, wo, vi, ctx, O.zero_states(cfg, B, np.float32))
            ref.append(O.crossentropy(p, vt)[0])
            lm.reset_states(B)
            lm.loss_acc.zero_()
            lm.forward_window(vi, ctx, vt, want_probs=False)
            got.append(lm.read_loss()[0])
        lm.set_states(carried)
        return float(np.mean(got)), float(np.mean(ref))

    curve = [(0,) + validate()]
    t0 = time.time()
    for step in range(steps):
        idx, tgt = text[:, step * T:(step + 1) * T], text[:, step * T + 1:(step + 1) * T + 1]
        masks = lm.draw_dropout_masks(B)
        om = [None] + [masks[l] for l in range(1, depth)]
        _, _, st = O.train_step(cfg, wo, opt, idx, ctx, tgt, st, om)
        lm.loss_acc.zero_()
        lm.train_window(idx, ctx, tgt, masks)
        lm.adam_step()
        if (step + 1) % every == 0:
            curve.append((step + 1,) + validate())
    print("convergence (%d steps, %.0f s): step, val loss HIP bf16, val loss f32 restatement" % (steps, time.time() - t0))
    for s, a, b in curve:
        print("  %4d  %.4f  %.4f  (%+.2f %%)" % (s, a, b, 100 * (a - b) / b))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r04_convergence_curve.json"), "w") as f:
            json.dump({"shape": {"depth": depth, "width": width, "length": T, "streams": B, "steps": steps},
                       "validation_loss": [{"step": s, "hip_bf16": a, "f32_restatement": b} for s, a, b in curve]}, f, indent=1)
    for s, a, b in curve:
        assert abs(a - b) < 0.02 * b, (s, a, b)
    (_, a0, b0), (_, a1, b1) = curve[0], curve[-1]
    assert abs(np.exp(a1) - np.exp(b1)) < 0.03 * np.exp(b1), (np.exp(a1), np.exp(b1))
    assert b1 < 0.8 * b0 and a1 < 0.8 * a0, curve          # (the chain has been learnt: from ~log(42) down towards its entropy rate)
