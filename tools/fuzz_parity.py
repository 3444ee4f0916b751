"""Randomised parity sweep (diagnostic): training gradients, loss, carried state and rating-window
probabilities of the HIP engine against the f64 oracle over random topologies and batch shapes."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from oracle import lstm_oracle as O
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
from tests.gradcheck import gradient_table



def run(n_cases, seed, verbose=True):
  rng = np.random.default_rng(seed)
  bad = 0
  for case in range(n_cases):
      depth = int(rng.integers(1, 5))
      width = int(rng.choice([32, 64, 96, 128, 160, 256, 512, 100, 50, 200, 333]))
      voc = int(rng.integers(5, 300))
      n_ctx = int(rng.integers(0, 3))
      B = int(rng.choice([1, 2, 3, 7, 8, 15, 16, 17, 24, 40, 100, 144, 200, 264, 512, 768, 1024, 1040, 1536]))
      T = int(rng.integers(1, 12))
      if B * T * width > 4e6:
          T = max(1, int(4e6 / (B * width)))
      use_masks = bool(rng.integers(0, 2))
      cfg = O.ModelConfig(depth, width, voc, n_ctx)
      w = O.init_weights(cfg, seed=int(rng.integers(1 << 30)), emb_std=0.3, dtype=np.float32)
      w64 = {k: v.astype(np.float64) for k, v in w.items()}
      lm = HipLM(depth, width, voc, n_ctx)
      lm.set_weights(w, hipabi.KL_PREC_BF16)
      lm.reset_states(B)
      idx = rng.integers(0, voc, (B, T))
      ctx = rng.integers(0, 200, (B, 1, max(n_ctx, 1))).repeat(T, axis=1)[:, :, :n_ctx]
      tgt = rng.integers(0, voc, (B, T))
      tgt[rng.random((B, T)) < 0.1] = -1
      st0 = [rng.standard_normal((B, width)) * 0.1 for _ in range(2 * depth)]
      lm.set_states(np.stack(st0, axis=1).astype(np.float32))
      masks = lm.draw_dropout_masks(B) if use_masks else None
      om = ([None] + [masks[l].astype(np.float64) for l in range(1, depth)]) if use_masks else None
      ref_p, ref_st, cache = O.forward_window(cfg, w64, idx, ctx, [s.astype(np.float64) for s in st0], om, keep_cache=True)
      ce, acc, _ = O.crossentropy(ref_p, tgt)
      g_data = O.backward_window(cfg, w64, idx, ctx, tgt, ref_p, cache, om, with_regularisers=False)
      lm.loss_acc.zero_()
      lm.train_window(idx, ctx, tgt, masks)
      l, a, r = lm.read_loss()
      # (E / Ctx*: the total AND the back-propagated part alone, tests/gradcheck.py)
      table = gradient_table(lm.layout, lm.get_grads(), g_data, O.regulariser_grads(cfg, w64))
      worst = max(max(e["max_over_maxnorm"], e.get("data_max_over_maxnorm", 0.0) if e.get("data_ref_norm", 0.0) > 0 else 0.0)
                  for e in table.values())
      st = lm.get_states()
      sterr = max(np.abs(st[:, k] - ref_st[k]).max() for k in range(2 * depth))
      # rating window in split precision from the same start
      lm.prepare(hipabi.KL_PREC_SPLIT)
      lm.set_states(np.stack(st0, axis=1).astype(np.float32))
      ref_i, _, _ = O.forward_window(cfg, w64, idx, ctx, [s.astype(np.float64) for s in st0])
      perr = np.abs(lm.forward_window(idx, ctx).cpu().numpy() - ref_i).max()
      ok = worst < 3e-2 and abs(l - ce) < 2e-2 * max(1, ce) and sterr < 3e-2 and perr < 3e-5
      bad += 0 if ok else 1
      if verbose or not ok:
        print(f"{'ok ' if ok else 'BAD'} L={depth} W={width} V={voc} C={n_ctx} B={B} T={T} masks={int(use_masks)}: "
              f"grad {worst:.1e} loss {abs(l - ce):.1e} state {sterr:.1e} probs {perr:.1e}", flush=True)
      del lm
  return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{bad} bad of {n}")
    sys.exit(1 if bad else 0)
